// CDNA4 (gfx950) kernels of the RRTMG_LW hot path.  Written for MI355X only.
//
// Mapping (DESIGN.md "Kernels"): the caller's arrays are (ncol, nlay) with the COLUMN index fastest
// (reference: src/rrtmg_lw_rad.nomcica.f90:219-276), so lanes = consecutive columns makes every load and
// store of profile, scratch and flux data a full-width coalesced access.  The work is split where the
// reference's data dependence allows it:
//
//   k_colsort  cloudy batches: the order in which the later kernels take the columns - by cloud top within windows of 256, where that
//              makes the sweeps' 64-column blocks alike (workspace arrays by position, the caller's arrays by column)   1 workgroup / window
//   k_colprep  per-column scalars: laytrop, precipitable water -> diffusivity secants, surface Planck
//              terms                                                    1 thread / column
//   k_cloudscan / k_cloudlay   cldprop, cloud optical depth per spectral band: the one quantity cldprop carries from layer
//              to layer per column, then the per-layer physics        1 thread / column, 1 thread / (column, layer)
//   k_cloudmc  cldprmc + the cloud set-up of rtrnmc (McICA)             1 thread / (column, layer)
//   k_blocksort  the 64-column blocks of a batch ordered by their highest cloudy layer, one hand-off level per group of blocks
//                                                                       1 workgroup / batch
//   k_layer    everything that is LOCAL to a layer: inatm + setcoef + taumol for all bands (the bands' absorption tables staged
//              in LDS, several bands per pass, for the 256 columns of a workgroup, which share the layer), then for each cell the
//              decision rtrn takes on its optical depth (series / table index) as a 4-byte code
//                                                                       1 thread / (column, layer)
//   k_sweepc / k_sweepz   the only vertically serial part: the down/up recurrences of rtrn / rtrnmr / rtrnmc over the cell codes
//              (transmittance table, the bands' Planck integrals and fractions in LDS); k_sweepc above the hand-off level of the
//              wave's block group and for cloud-free calls (1 thread / (column, band)), k_sweepz in the cloud zone (1 thread /
//              (column, quad)); a workgroup = the bands of one group, flux partials added over the group in LDS
//   k_flux     group partials -> fluxes, net fluxes, heating rates           1 thread / (column, level)
//   k_subcol_* McICA sub-column generator (bit masks), k_alpha           see the section below
//
// Reference lines are cited per routine.  No CPU fallback exists anywhere in this file.
#include <hip/hip_runtime.h>
#include "kissjump.hpp"

// ------------------------------------------------------------------------------------------------
// Tuning switches.  This translation unit carries one-source measurement switches (-DRRLW_...: geometry of the kernels, numerics variants,
// two knock-outs that give WRONG results); the shipped libraries are built with none of them (only -DRRLW_G256 for the 256-g-point
// library).  What a library was built with is recorded in its build-flags word (rrtmg_lw_hip_build_flags): rrtmg_lw_hip_init refuses a
// library whose results differ from the product's (knock-out or numerics variant) unless RRTMG_LW_ALLOW_TUNE_BUILD=1, and a knock-out
// does not compile outside a tuning build (-DRRLW_TUNE: the benchmark's kernels only).
// ------------------------------------------------------------------------------------------------
#if (defined(RRLW_KO_LDS_UNIFORM) || defined(RRLW_KO_HALFG) || defined(RRLW_KO_STORES) || defined(RRLW_KO_HALFSTORE)) && !defined(RRLW_TUNE)
#error "RRLW_KO_* knock-outs give wrong results: tuning builds (-DRRLW_TUNE) only"
#endif
#ifdef RRLW_TUNE
#define RRLW_BF_TUNE 1u
#else
#define RRLW_BF_TUNE 0u
#endif
#if defined(RRLW_KO_LDS_UNIFORM) || defined(RRLW_KO_HALFG) || defined(RRLW_KO_STORES) || defined(RRLW_KO_HALFSTORE)
#define RRLW_BF_KNOCKOUT 2u
#else
#define RRLW_BF_KNOCKOUT 0u
#endif
#if (defined(RRLW_CODE_BITS) && RRLW_CODE_BITS != 32) || defined(RRLW_DECODE_F64) || defined(RRLW_SWEEP_EXPF) || defined(RRLW_EXACT_DIV) || defined(RRLW_FDIV_TWO_NEWTON)
#define RRLW_BF_NUMERICS 4u
#else
#define RRLW_BF_NUMERICS 0u
#endif
#ifdef RRLW_G256
#define RRLW_BF_G256 8u
#else
#define RRLW_BF_G256 0u
#endif
// kernel-geometry switches given on the command line (same results, other speed)
#if defined(RRLW_LAYER_BLOCK) || defined(RRLW_STAGE_DOUBLES) || defined(RRLW_LAYER_WAVES) || defined(RRLW_LOAD_CHUNK) || defined(RRLW_CLOUD_QUADS) || \
    defined(RRLW_MINOR_WIN) || defined(RRLW_COLSORT_WIN) || defined(RRLW_LAYER_PASS_PER_BAND) || defined(RRLW_LAYER_PASSES_3) || defined(RRLW_LAYER_PASSES_1) || \
    defined(RRLW_LAYER_SYNCTHREADS) || defined(RRLW_LAYER_STAMPS) || defined(RRLW_NO_NT) || defined(RRLW_SWEEPC_P0) || defined(RRLW_SWEEPC_P2D) || \
    defined(RRLW_SWEEPC_WAVES_CAP) || defined(RRLW_SWEEPC_QUAD_BARRIER) || defined(RRLW_SWEEPC_CODES) || defined(RRLW_SWEEPC_SPLIT) || defined(RRLW_SWEEPC_SPLIT3) || \
    defined(RRLW_SWEEPC_SPLIT_P1) || defined(RRLW_SWEEPZ_CT_SLOTS) || defined(RRLW_SWEEPZ_G2) || defined(RRLW_SWEEPZ_WAVES_G2) || defined(RRLW_SWEEPZ_WAVES_G1) || \
    defined(RRLW_SWEEPZ_WAVES_IDRV) || defined(RRLW_GEN_BESIDE_SWEEP) || defined(RRLW_FANOUT_MAX) || defined(RRLW_KI_SALU) || \
    defined(RRLW_DBG_DUMP)      /* (driver.hip: one more entry, rrtmg_lw_hip_debug_scratch - reads k_layer's cell codes back; tuning builds only) */
#define RRLW_BF_GEOMETRY 16u
#else
#define RRLW_BF_GEOMETRY 0u
#endif

#include <algorithm>
#include <type_traits>
#include <utility>

#include "tables.hpp"

namespace rrlw {

constexpr unsigned BUILD_FLAGS = RRLW_BF_TUNE | RRLW_BF_KNOCKOUT | RRLW_BF_NUMERICS | RRLW_BF_G256 | RRLW_BF_GEOMETRY;

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

// setcoef quantities of one (layer, column) - indices into LayerCoef::f
enum Field {
    F_FAC00, F_FAC01, F_FAC10, F_FAC11,
    F_COLH2O, F_COLCO2, F_COLO3, F_COLN2O, F_COLCO, F_COLCH4, F_COLO2, F_COLBRD,
    F_COLDRY, F_SELFFAC, F_SELFFRAC, F_FORFAC, F_FORFRAC, F_MINORFRAC, F_SCALEMINOR, F_SCALEMINORN2,
    F_PAVEL, F_WX1, F_WX2, F_WX3, F_WX4,
    NFIELD
};
// per-column doubles, each [ncolb]
enum PerCol { PC_PLANKBND = 0, PC_DPLANKBND = 16, PC_SECDIFF = 32, NPERCOL = 48 };
// per-cell transmittance codes handed from k_layer to the sweeps, each [NQUAD][nlay][ncolb][4] (see cell_code):
// gas optical depth of every cell, total (gas + cloud) optical depth of the cells of cloudy layers
enum Scr { S_CODE, S_CODET, NSCR };
// per-band partial fluxes, each [band][level][column] of {total-sky, clear-sky}: downward, upward, d(upward)/dT
struct alignas(16) Part2 { double a, b; };

// the per-cell codes handed from k_layer to the sweeps (all arithmetic stays float64): in registers a float (scr_t, cell_code), in
// memory RRLW_CODE_BITS bits per cell, four cells (a quad) per record.  32 = the float itself; 24 and 16 are measurement variants that
// keep the table index whole and round the series-branch optical depth to 18 / 11 mantissa bits (pack4 / unpack4).
typedef float scr_t;
#ifndef RRLW_CODE_BITS
#define RRLW_CODE_BITS 32
#endif
constexpr int CODE_BITS = RRLW_CODE_BITS;
static_assert(CODE_BITS == 32 || CODE_BITS == 24 || CODE_BITS == 16, "bits per cell code");
constexpr int CODE_WORDS = CODE_BITS / 8;        // 32-bit words per quad record
constexpr int CODE_BYTES = 4 * CODE_WORDS;

struct Workspace {
    int ncolb;          // column stride (batch capacity)
    int nlay;
    double *percol;     // [NPERCOL][ncolb]
    int *laytrop;       // [ncolb]
    int *ncbands;       // [ncolb]
    double *odcld;      // [16][nlay][ncolb]   secdiff(ib) * taucloud
    double *efcl;       // [16][nlay][ncolb]   rtrn: (1 - exp(-odcld)) * cldfrac
    int *cflag;         // [nlay+2][ncolb]  bit0 icldlyr, bit1 istcldd (first cloudy level of a block, downward sweep), bit2 istcld (upward); cflag[0] bit3 = column has cloud
    double2 *ovl;       // rtrnmr's overlap factors of the cloudy levels: [2: down, up][nlay+1][3][ncolb] {facclr1, faccld1} {faccmb1, faccmb2} {facclr2, faccld2}
    // Where the clouds end.  A sweep wave works on a BLOCK of 64 consecutive columns; above the highest cloud of its block every layer is
    // clear in every column and the cheap clear-sky sweep (k_sweepc) serves.  k_cloudscan / k_cloudmc record the top per block, k_blocksort
    // orders the blocks by it (deepest first) and gives every group of SORT_GROUP consecutive sorted blocks ONE hand-off level, the
    // group's highest top: a sweep workgroup takes 1, 2, 3, 4, 6 or 12 consecutive sorted blocks of one group, whatever the kernel.
    int *btop;          // [nblk]  highest layer that holds cloud in any column of the block (0: none)
    int *order;         // [nslot] sorted position -> block (positions past the last block: nblk, i.e. columns past the end)
    int *hgrp;          // [nslot / SORT_GROUP] hand-off level of the group
    int *hblk;          // [nblk]  hand-off level of the block's group (k_flux)
    int *bbot;          // [nblk]  lowest layer that holds cloud in any column of the block (nlay + 1: none)
    int *hbot;          // [nslot / SORT_GROUP] lowest cloudy layer of the group: below it the cloud-zone sweep runs its clear-sky body without the cloudy-level inputs
    double *hand;       // [5][NQUAD][ncolb][4] radiances handed from sweep to sweep at the block's hand-off level: 0 downward (k_sweepc<.,1> -> k_sweepz), 1 / 2 upward
                        // total / clear (k_sweepz -> k_sweepc<.,2>), 3 / 4 their d/dT (idrv = 1)
    unsigned *scr[NSCR];   // [NQUAD][nlay][ncolb] quad records of CODE_WORDS words
    unsigned *fw;       // [NFW][nlay][ncolb]   binary-key bands: Planck-fraction interpolation (js << 28 | 28-bit fs)
    // k_sweepc's partials, summed over the bands of a GROUP (bands with the same number of quads, swept by one workgroup and added in
    // LDS): [NGROUP_MAX][nlay+1][ncolb].  gdn1 / gup1: 8 bytes where the clear-sky stream equals the total one (downward at and above the
    // batch's highest cloud, both directions of a cloud-free call); gup / gdp {total, clear}: upward above the clouds, d(flux)/dT
    double *gdn1, *gup1;
    Part2 *gup, *gdp, *gdn;     // gdn {total, clear}: downward inside the cloud zone (k_sweepz)
    // column stride of those slabs: ncolb - or, for a batch too small to fill the chip, whose bands are swept ONE per workgroup and leave a
    // slab each (SweepArgs::split), the batch's own width (the slabs' memory then holds sixteen narrow slabs instead of four wide ones)
    int pcb;
    int *err;           // [1] first physics error code
    // McICA (rtrnmc): per-g-point cloud terms, written by k_cloudmc
    double *odg;        // [NQUAD][nlay][ncolb][4]     secdiff(band) * taucmc(g)
    float *cfef;        // [NQUAD][nlay][ncolb][8]     {cldfmc(g) x4, efclfrac(g) x4}
    unsigned *mask;     // [MASK_WORDS][nlay][mask_stride]      sub-column cloud mask of ALL columns of the call (indexed by the global
    size_t mask_stride; //                             column mask_col0 + col0 + col), bit k of word w = sub-column 32 w + k
    size_t mask_col0;
    // The order in which the batch's columns are taken (k_colsort): position (slot) -> column of the batch, or null = as they come.  Every
    // workspace array above is indexed by SLOT, the caller's arrays by COLUMN (pcol).
    // k_layer's (window, layer) pairs whose cells do not fit the narrow staging window, left by the narrow launch for the wide one:
    // {count, unused, pair ..}; null: no wide launch follows (every workgroup keeps the narrow window)
    int *wide;          // [2 + windows * nlay]
    int *perm;          // [ncolb rounded up to whole windows]
    int *wsort;         // [windows] 1: the window's columns are taken in another order than they lie
    // ... and for such windows the rows the sweeps read of the caller's arrays at EVERY level - layer and interface temperatures, cloud
    // fraction - once more in position order (k_colsort): [nlay | nlay+1 | nlay][ncolb]
    double *tlayc, *tlevc, *cldfc;
};
#ifndef RRLW_COLSORT_WIN
#define RRLW_COLSORT_WIN 256
#endif
constexpr int COLSORT_WIN = RRLW_COLSORT_WIN;       // columns of a k_colsort window
// (a wavefront's positions lie in ONE window - 64 consecutive positions from a multiple of 64, or all the last column's: the window's flag is
// a scalar load, and a window that kept its order does not wait for its `perm` entries, a cold vector load at the head of every kernel)
__device__ __forceinline__ bool pmoved(const Workspace &W, int slot) { return W.perm && W.wsort[__builtin_amdgcn_readfirstlane(slot) / COLSORT_WIN] != 0; }
__device__ __forceinline__ int pcol(const Workspace &W, int slot)
{
    if (!W.perm) return slot;
    const int moved = W.wsort[__builtin_amdgcn_readfirstlane(slot) / COLSORT_WIN];     // (uniform address: a scalar load)
    return moved ? W.perm[slot] : slot;
}
// One value of a row of the caller's (column-fastest) arrays: wave-uniform row pointer in a buffer descriptor, the lane's column as a 32-bit
// byte offset - the column is a loaded value (pcol), and as part of a 64-bit address it would cost every load its own address arithmetic.
__device__ __forceinline__ double col_load(const double *row, unsigned off8)
{
    typedef unsigned int u32x2c __attribute__((ext_vector_type(2)));
    const u32x2c v = __builtin_amdgcn_raw_buffer_load_b64(__builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(row), 0, 0x7ffffff0, 0x00020000), (int)off8, 0, 0);
    double r;
    __builtin_memcpy(&r, &v, 8);
    return r;
}

// GCM-interface inputs (device pointers, column stride = ncol_total), reference src/rrtmg_lw_rad.nomcica.f90:219-276
struct GcmIn {
    const double *play, *plev, *tlay, *tlev, *tsfc, *h2ovmr, *o3vmr, *co2vmr, *ch4vmr, *n2ovmr, *o2vmr;
    const double *cfc11vmr, *cfc12vmr, *cfc22vmr, *ccl4vmr, *emis;
    const double *cldfr, *taucld, *cicewp, *cliqwp, *reice, *reliq, *tauaer;
    const double *tauctot = nullptr;    // (ncol,nlay), optional: the sum of taucld over the bands where the caller has formed it (host entry, inflglw >= 1); taucld is null then
};
// prepared-column inputs (device pointers, column stride = ncol_total), reference src/rrtmg_lw.1col.f90:497-580
struct ColIn {
    const double *pavel, *tavel, *pz, *tz, *tbound, *semiss, *coldry, *wkl, *wbrodl, *wx, *pwvcm;
    const double *cldfrac, *tauc, *ciwp, *clwp, *rei, *rel, *taua;
};
// McICA cloud inputs (device pointers), reference src/rrtmg_lw_rad.f90:267-298
struct McIn {
    const double *cldfmcl, *taucmcl, *ciwpmcl, *clwpmcl;   // (140, ncol_total, nlay)
    const double *reicmcl, *relqmcl;                       // (ncol_total, nlay)
};
struct FluxOut {
    double *uflx, *dflx, *hr, *uflxc, *dflxc, *hrc, *duflx_dt, *duflxc_dt;   // stride ncol_total
    double *fnet, *fnetc;                                                     // optional (column entry)
};

enum ErrCode { E_NONE = 0, E_ICE_SMALL = 1, E_ICE_BOUNDS = 2, E_ICE_GEN_BOUNDS = 3, E_LIQ_BOUNDS = 4, E_BAD_FLAG = 5,
               E_MC_INFLAG1 = 6, E_KISS_PMID = 7 };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// a / b for normal, finite operands in k_layer: hardware reciprocal seed (2^-24), ONE Newton step (2^-48), one residual correction
// q + (a - b q) r, whose error is the product of the two (6 VALU operations in one dependency chain instead of the ~13 of the IEEE
// sequence with its scale / fix-up steps; a third of k_layer's VALU instructions were divisions).  The quotient is within an ulp of
// the correctly rounded one; where it feeds an index (`int(...)`, the 1e-4-quantised transmittance table) a different index needs the
// exact quotient to sit within ~1e-16 of a boundary: tools/divtest.hip finds 0 quotients and 0 table indices different from the IEEE
// division in 2.1e9 operands, with one Newton step as with two (profiles/round2_divtest.txt; round 3 made one the default: -0.3 ms).
// -DRRLW_EXACT_DIV restores the IEEE division, -DRRLW_FDIV_TWO_NEWTON the second step.
__device__ __forceinline__ double fdiv(double a, double b)
{
#ifdef RRLW_EXACT_DIV
    return a / b;
#else
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
#ifdef RRLW_FDIV_TWO_NEWTON
    r = fma(fma(-b, r, 1.0), r, r);
#endif
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
#endif
}

// ------------------------------------------------------------------------------------------------
// k_colsort : the order in which every later kernel takes the batch's columns.  The sweeps decide per wavefront - 64 consecutive positions -
//             where the clouds end (Workspace::btop): one deep tower among 64 shallow columns sends all of them through the cloud-zone sweep
//             up to its top, and one cloudy column among clear ones likewise.  Within each window of COLSORT_WIN consecutive columns (the
//             columns of ONE k_layer workgroup, so that what a workgroup reads of the caller's arrays is the same few cache lines in any
//             order) the columns are ordered by their highest cloudy layer, deepest first, equal tops in column order - a homogeneous window
//             keeps its order.  Results do not depend on the order (a column does not depend on its neighbours); the reference has no
//             counterpart: it takes the columns one by one (src/rrtmg_lw_rad.nomcica.f90:472).
// ------------------------------------------------------------------------------------------------
// A window is reordered only where that pays: reading the caller's arrays out of order costs every kernel something (eight-byte accesses
// spread over the window's cache lines instead of consecutive ones), so the order is taken when it removes at least `min_gain` block-levels
// from the cloud zone - the sum over the window's four 64-column blocks of the block's highest cloudy layer, as the columns lie against
// sorted - and the window keeps its order otherwise (a homogeneous deck, a tower system that fills whole blocks).
constexpr int COLSORT_TY = 4;       // threads per column: the walk over the layers and the copy of the rows in four parts
template <bool GCM>
__global__ __launch_bounds__(COLSORT_WIN * COLSORT_TY) void k_colsort(Workspace W, GcmIn g, ColIn c, int ncol, int col0, int nct, int min_gain)
{
    constexpr int NB = COLSORT_WIN / 64;
    __shared__ int s_key[COLSORT_WIN], s_perm[COLSORT_WIN], s_top[COLSORT_TY][COLSORT_WIN];
    __shared__ int s_nat[NB], s_srt[NB];
    const int t = threadIdx.x, ty = threadIdx.y, w0 = blockIdx.x * COLSORT_WIN, col = w0 + t;
    const int nlay = W.nlay;
    int top = -1;                                   // a position past the last column: after everything
    if (col < ncol) {
        const double *cf = (GCM ? g.cldfr : c.cldfrac) + (size_t)col0 + col;
        const int per = (nlay + COLSORT_TY - 1) / COLSORT_TY, l0 = ty * per + 1, l1 = min(nlay, l0 + per - 1);
        top = 0;
#pragma unroll 8
        for (int lay = l0; lay <= l1; lay++) { if (cf[(size_t)nct * (lay - 1)] >= 1.e-6) top = lay; }      // (k_cloudscan's `cloudy`)
    }
    s_top[ty][t] = top;
    __syncthreads();
    int rank = 0;
    if (ty == 0) {
#pragma unroll
        for (int k = 1; k < COLSORT_TY; k++) top = max(top, s_top[k][t]);
        s_key[t] = (nlay - top) * COLSORT_WIN + t;  // deepest first, then column order
        int mx = max(top, 0);                       // the block's top as the columns lie: a wave = 64 consecutive columns
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
        if ((t & 63) == 0) s_nat[t >> 6] = mx;
    }
    __syncthreads();
    if (ty == 0) {
        const int key = s_key[t];
#pragma unroll 8
        for (int i = 0; i < COLSORT_WIN; i++) rank += s_key[i] < key ? 1 : 0;
        if ((rank & 63) == 0) s_srt[rank >> 6] = max(top, 0);      // ... and sorted: the first of every 64 is the deepest
    }
    __syncthreads();
    int gain = 0;
#pragma unroll
    for (int b = 0; b < NB; b++) gain += s_nat[b] - s_srt[b];
    const bool keep = gain * 4 < min_gain * NB;     // (uniform over the workgroup; min_gain is quoted for four blocks)
    if (ty == 0) {
        if (keep) rank = t;
        if (t == 0) W.wsort[blockIdx.x] = keep ? 0 : 1;
        W.perm[w0 + rank] = col;
        s_perm[rank] = col;
    }
    if (keep) return;
    // A reordered window: the rows of the caller's arrays that the sweeps read at EVERY level - layer and interface temperatures, cloud
    // fraction - once more in position order (Workspace::tlayc ..), read through the new order, written consecutively.  (Out-of-order
    // eight-byte reads cost the sweeps 5 ms per 1e6 deep-cloud columns where they recurred per level and band.)
    if constexpr (GCM) {
        __syncthreads();
        const int slot = w0 + t;
        if (slot < ncol) {
            const size_t g0 = (size_t)col0 + s_perm[t];
#pragma unroll 4
            for (int lev = ty; lev <= nlay; lev += COLSORT_TY) {         // level 0 .. nlay; layer lev + 1 for the two layer arrays
                const size_t gi = g0 + (size_t)nct * lev, o = (size_t)lev * W.ncolb + slot;
                W.tlevc[o] = g.tlev[gi];
                if (lev < nlay) { W.tlayc[o] = g.tlay[gi]; W.cldfc[o] = g.cldfr[gi]; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_colprep : the per-column reductions of inatm (src/rrtmg_lw_rad.nomcica.f90:785-870: amttl, wvttl ->
//             pwvcm), setcoef's laytrop count and surface Planck terms (src/rrtmg_lw_setcoef.f90:173-215,
//             :312-313) and the diffusivity secants (src/rrtmg_lw_rtrn.f90:265-288).
// ------------------------------------------------------------------------------------------------
template <bool GCM>
__global__ __launch_bounds__(256) void k_colprep(DevTables T, Workspace W, GcmIn g, ColIn c, int ncol, int col0, int nct, int idrv, int istart, int init_cloud)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    const size_t gc = (size_t)col0 + pcol(W, col);
    const int nlay = W.nlay;
    const double *S = T.stat;
    const double *totplnk = S + T.sl.totplnk, *totplnkd = S + T.sl.totplnkderiv;
    const double *totplk16 = S + T.sl.totplk16, *totplk16d = S + T.sl.totplk16deriv;
    const double amd = 28.9660, amw = 18.0160, avogad = 6.02214199e+23, grav = 9.8066;

    const double tbound = GCM ? g.tsfc[gc] : c.tbound[gc];
    const double pz0 = GCM ? g.plev[gc] : c.pz[gc];
    const int indbound = clampi((int)(tbound - 159.), 1, 180);
    const double tbndfrac = tbound - 159. - (double)indbound;
#pragma unroll
    for (int b = 0; b < NBND; b++) {
        const double semiss = GCM ? g.emis[gc + (size_t)nct * b] : c.semiss[gc + (size_t)nct * b];
        const double *tp = (b == 15 && istart == 16) ? totplk16 : totplnk + 181 * b;        // :233-242
        const double *td = (b == 15 && istart == 16) ? totplk16d : totplnkd + 181 * b;
        const double pb = semiss * (tp[indbound - 1] + tbndfrac * (tp[indbound] - tp[indbound - 1]));
        double dpb = 0.0;
        if (idrv == 1) dpb = semiss * (td[indbound - 1] + tbndfrac * (td[indbound] - td[indbound - 1]));
        W.percol[(size_t)(PC_PLANKBND + b) * W.ncolb + col] = pb;
        W.percol[(size_t)(PC_DPLANKBND + b) * W.ncolb + col] = dpb;
    }
    double amttl = 0.0, wvttl = 0.0, pzlo = pz0;
    int laytrop = 0;
    // (a thread walks its column alone: unrolled, the loads of eight layers are in flight together instead of one round trip per layer -
    // on the critical path of every call that is a single batch)
#pragma unroll 8
    for (int lay = 1; lay <= nlay; lay++) {
        const size_t gi = gc + (size_t)nct * (lay - 1);
        const double pavel = GCM ? g.play[gi] : c.pavel[gi];
        if (!(log(pavel) <= 4.56)) laytrop++;
        if (GCM) {
            const double w1v = g.h2ovmr[gi];
            const double pzl = g.plev[gc + (size_t)nct * lay];
            const double amm = (1. - w1v) * amd + w1v * amw;
            const double coldry = (pzlo - pzl) * 1.e3 * avogad / (1.e2 * grav * amm * (1. + w1v));
            pzlo = pzl;
            const double w1 = coldry * w1v;
            amttl = amttl + coldry + w1;
            wvttl = wvttl + w1;
        }
    }
    W.laytrop[col] = laytrop;
    double pwvcm;
    if (GCM) {
        const double wvsh = (amw * wvttl) / (amd * amttl);
        pwvcm = wvsh * (1.e3 * pz0) / (1.e2 * grav);
    } else {
        pwvcm = c.pwvcm[gc];
    }
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
    // (the cloud-free defaults, and the block tops k_cloudmc raises with atomics; k_cloudscan writes all of these itself and may run BESIDE
    // this kernel - run_prep - so they are left alone for it: init_cloud = 0)
    if (init_cloud) {
        W.ncbands[col] = 1;
        W.cflag[col] = 0;
        if ((col & 63) == 0) { W.btop[col >> 6] = 0; W.bbot[col >> 6] = nlay + 1; }
    }
}

// ------------------------------------------------------------------------------------------------
// k_cloud : cldprop (src/rrtmg_lw_cldprop.f90:50-295), cloud optical depth along the diffusivity
//           angle (rtrn :321-334 / rtrnmr :333-343).  mode 1 = rtrn, 2 = rtrnmr.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int icb_map(int ib, int ind)   // cldprop :167-169 (1-based ib)
{
    if (ind == 0) return 1;
    if (ind == 2) return ib;
    return ib <= 2 ? ib : (ib <= 5 ? 3 : (ib <= 8 ? 4 : 5));
}

// cldprop carries two things from layer to layer of a column (src/rrtmg_lw_cldprop.f90:173-293): `ncbands`, assigned by the layers that
// hold ice or liquid water and left alone by the others, and nothing else that matters (iceind / liqind are reassigned before every
// use).  k_cloudscan walks a column once and records the band count in effect at each layer; k_cloudlay then does the per-layer physics
// for all (column, layer) pairs in parallel - the serial kernel this replaces kept one wave per 64 columns busy for 0.8 ms per batch.
__device__ __forceinline__ bool cloud_layer_enters(double cf, double cwp, double tauctot)      // cldprop :186
{
    const double cldmin = 1.e-20;
    return cf >= cldmin && (cwp >= cldmin || tauctot >= cldmin);
}
template <bool GCM>
__device__ __forceinline__ double cloud_tauctot(const GcmIn &g, const ColIn &c, size_t gc, size_t gi, int nct, int lay)
{
    if constexpr (GCM) { if (g.tauctot) return g.tauctot[gi]; }
    double tauctot = 0.0;                               // taucld (16,ncol,nlay) | tauc (ncol,16,nlayers)
    for (int ib = 0; ib < NBND; ib++)
        tauctot = tauctot + (GCM ? g.taucld[ib + (size_t)NBND * gi] : c.tauc[gc + (size_t)nct * (ib + (size_t)NBND * (lay - 1))]);
    return tauctot;
}

// Maximum-random overlap factors of ONE cloudy level, src/rrtmg_lw_rtrnmr.f90:347-506 (upward loop :347-425, downward loop :427-506;
// the two loops are mirror images).  cl = cloud fraction of the level, cn = of the next level in sweep direction, cp = of the previous
// one (read only when that level is cloudy, i.e. !first), last = the level is the last of the sweep.  rat1 / rat2 carry from cloudy
// level to cloudy level as in the reference.  faccmb1/2, which the reference reads uninitialised when first, are ZERO.
struct OvlFac { double clr1, cld1, cmb1, cmb2, clr2, cld2; bool rat1, rat2; };     // rat1 / rat2 (0 or 1 in the reference): carried to the next cloudy level
__device__ __forceinline__ OvlFac mr_step(double cl, double cn, double cp, bool first, bool last, bool brat1, bool brat2)
{
    const double rat1 = brat1 ? 1. : 0., rat2 = brat2 ? 1. : 0.;
    // (everything by value: carried state behind references ends up in scratch memory).  Of the reference's seven ways through this
    // block at most one needs a quotient, so the ways are told apart by predicates, ONE division runs on the selected operands and the
    // result is routed by selects: no divergent branches (the nested form cost ~200 instructions per cloudy level, this one ~70).
    const bool act = !last, up = cn >= cl;
    const double fmx = fmax(cl, cp), fmn = fmin(cl, cp);
    const bool u_f = act && up && first && cl < 1.;                 // facclr2 = (cn - cl) / (1 - cl)
    const bool u_gt = act && up && !first && cn > fmx;              // facclr1 = rat2, facclr2 = (cn - fmx) / (1 - fmx)
    const bool u_lt = act && up && !first && cn < fmx;              // facclr1 = (cn - cl) / (cp - cl)
    const bool u_eq = act && up && !first && !(cn > fmx) && !(cn < fmx);      // facclr1 = rat2
    const bool d_f = act && !up && first;                           // faccld2 = (cl - cn) / cl
    const bool d_le = act && !up && !first && cn <= fmn;            // faccld1 = rat1, faccld2 = (fmn - cn) / fmn
    const bool d_gt = act && !up && !first && !(cn <= fmn);         // faccld1 = (cl - cn) / (cl - fmn)
    double num = 0.0, den = 1.0;
    num = u_f ? cn - cl : num;     den = u_f ? 1. - cl : den;
    num = u_gt ? cn - fmx : num;   den = u_gt ? 1. - fmx : den;
    num = u_lt ? cn - cl : num;    den = u_lt ? cp - cl : den;
    num = d_f ? cl - cn : num;     den = d_f ? cl : den;
    num = d_le ? fmn - cn : num;   den = d_le ? fmn : den;
    num = d_gt ? cl - cn : num;    den = d_gt ? cl - fmn : den;
    const double q = fdiv(num, den);
    const double clr1 = (u_gt || u_eq) ? rat2 : (u_lt ? q : 0.0);
    const double clr2 = (u_f || u_gt) ? q : 0.0;
    const double cld1 = d_le ? rat1 : (d_gt ? q : 0.0);
    const double cld2 = (d_f || d_le) ? q : 0.0;
    const bool r1 = (act && up) ? (clr1 > 0. || clr2 > 0.) : (act ? false : brat1);
    const bool r2 = (act && !up) ? (cld1 > 0. || cld2 > 0.) : (act ? false : brat2);
    double cmb1 = 0.0, cmb2 = 0.0;
    if (!first) {
        const double cx = last ? 0.0 : cn;          // beyond the last level the reference's neighbour fraction is taken as 0
        cmb1 = fmax(0., fmin(cx - cl, cp - cl));
        cmb2 = fmax(0., fmin(cl - cx, cl - cp));
    }
    return OvlFac{clr1, cld1, cmb1, cmb2, clr2, cld2, r1, r2};
}

// cflag[lay][col]: bit 0 = the layer holds cloud for the sweeps (cldfrac >= 1e-6), bits 8.. = cldprop's ncbands in effect at the layer;
// cflag[0][col] bit 3 = the column holds cloud.  mode 2 (rtrnmr): the overlap factors of every cloudy level for both sweep directions
// (the reference's two set-up loops, rtrnmr :347-506) go to W.ovl, "first cloudy level of a block" to bits 1 (downward) and 2 (upward).
template <bool GCM>
__global__ __launch_bounds__(256) void k_cloudscan(Workspace W, GcmIn g, ColIn c, int ncol, int col0, int nct, int inflag, int iceflag, int liqflag, int mode)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int nlay = W.nlay;
    int top = 0, bot = nlay + 1;
    if (col < ncol) {
    const size_t gc = (size_t)col0 + pcol(W, col);
    const double *cldfr = GCM ? g.cldfr : c.cldfrac;
    const double *ciwp_ = GCM ? g.cicewp : c.ciwp, *clwp_ = GCM ? g.cliqwp : c.clwp;
    int ncbands = 1, anycloud = 0;
    const size_t ncb = W.ncolb;
    bool prevcld = false, rat1 = false, rat2 = false;       // upward sweep order = this loop's order
    double cfprev = 0.0;
#pragma unroll 4
    for (int lay = 1; lay <= nlay; lay++) {
        const size_t gi = gc + (size_t)nct * (lay - 1);
        const double cf = cldfr[gi], ciwp = ciwp_[gi], clwp = clwp_[gi];
        const double cwp = ciwp + clwp;
        bool enters = cf >= 1.e-20 && cwp >= 1.e-20;
        if (cf >= 1.e-20 && !enters) enters = cloud_layer_enters(cf, cwp, cloud_tauctot<GCM>(g, c, gc, gi, nct, lay));
        if (enters) {
            if (inflag == 0 || inflag == 1) ncbands = 16;
            else if (inflag == 2) {
                if (ciwp != 0.0) { if (iceflag == 1) ncbands = 5; else if (iceflag == 2 || iceflag == 3) ncbands = 16; }
                if (clwp != 0.0 && liqflag == 1) ncbands = 16;
            }
        }
        const int cloudy = cf >= 1.e-6;
        if (cloudy && !anycloud) bot = lay;
        anycloud |= cloudy;
        if (cloudy) top = lay;
        int first = 0;
        if (mode == 2) {
            if (cloudy) {
                first = !prevcld;
                const OvlFac mf = mr_step(cf, lay < nlay ? cldfr[gi + nct] : 0.0, cfprev, first, lay == nlay, rat1, rat2);
                rat1 = mf.rat1; rat2 = mf.rat2;
                prevcld = true;
                cfprev = cf;
                double2 *o = W.ovl + ((size_t)((nlay + 1) + lay) * 3) * ncb + col;
                o[0] = make_double2(mf.clr1, mf.cld1); o[ncb] = make_double2(mf.cmb1, mf.cmb2); o[2 * ncb] = make_double2(mf.clr2, mf.cld2);
            } else prevcld = false;
        }
        W.cflag[(size_t)lay * ncb + col] = cloudy | (first << 2) | (ncbands << 8);
    }
    if (mode == 2 && anycloud) {            // downward sweep order
        prevcld = false; rat1 = false; rat2 = false; cfprev = 0.0;
        for (int lay = top; lay >= 1; lay--) {
            const size_t gi = gc + (size_t)nct * (lay - 1);
            const double cf = cldfr[gi];
            if (cf >= 1.e-6) {
                const bool first = !prevcld;
                const OvlFac mf = mr_step(cf, lay > 1 ? cldfr[gi - nct] : 0.0, cfprev, first, lay == 1, rat1, rat2);
                rat1 = mf.rat1; rat2 = mf.rat2;
                prevcld = true;
                cfprev = cf;
                double2 *o = W.ovl + ((size_t)lay * 3) * ncb + col;
                o[0] = make_double2(mf.clr1, mf.cld1); o[ncb] = make_double2(mf.cmb1, mf.cmb2); o[2 * ncb] = make_double2(mf.clr2, mf.cld2);
                if (first) W.cflag[(size_t)lay * ncb + col] |= 2;
            } else prevcld = false;
        }
    }
    W.ncbands[col] = ncbands;
    W.cflag[col] = anycloud ? 8 : 0;
    W.cflag[(size_t)(nlay + 1) * W.ncolb + col] = 0;
    }
    // highest and lowest cloudy layer of the block: the 64 columns of a block are one workgroup (one wave) of this kernel (a cloud-free
    // column holds 0 / nlay + 1); written, not raised with atomics: nothing has to initialise them in front of this kernel
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { top = max(top, __shfl_xor(top, o, 64)); bot = min(bot, __shfl_xor(bot, o, 64)); }
    if (threadIdx.x == 0 && col < ncol) { W.btop[col >> 6] = top; W.bbot[col >> 6] = bot; }
}

// k_cloudlay : cldprop for one (column, layer) (src/rrtmg_lw_cldprop.f90:173-293) and the cloud optical depth along the diffusivity
//              angle per SPECTRAL band (rtrn :321-349 / rtrnmr :333-343).  mode 1 = rtrn (also the effective emissivity term), 2 = rtrnmr.
template <bool GCM>
__global__ __launch_bounds__(256) void k_cloudlay(DevTables T, Workspace W, GcmIn g, ColIn c, int ncol, int col0, int nct, int mode,
                                                  int inflag, int iceflag, int liqflag)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    const int lay = blockIdx.y + 1;
    const size_t gc = (size_t)col0 + pcol(W, col);
    const int nlay = W.nlay;
    const size_t ncb = W.ncolb;
    const size_t gi = gc + (size_t)nct * (lay - 1);
    const double *S = T.stat;
    const double *absice1 = S + T.sl.absice1, *absice2 = S + T.sl.absice2, *absice3 = S + T.sl.absice3, *absliq1 = S + T.sl.absliq1;
    const double cf = (GCM ? g.cldfr : c.cldfrac)[gi];
    const double ciwp = (GCM ? g.cicewp : c.ciwp)[gi], clwp = (GCM ? g.cliqwp : c.clwp)[gi];
    const int flagword = W.cflag[(size_t)lay * ncb + col];
    const int cloudy = flagword & 1, nb_at = flagword >> 8;       // the band count cldprop had reached when it worked on this layer
    const int ncbands = W.ncbands[col];                            // ... and at the end of the column: rtrn's band -> cloud-band map (:343-349)
    const double cwp = ciwp + clwp;
    bool enters = cf >= 1.e-20 && cwp >= 1.e-20;
    if (cf >= 1.e-20 && !enters) enters = cloud_layer_enters(cf, cwp, cloud_tauctot<GCM>(g, c, gc, gi, nct, lay));
    // inflag 2: how the ice / liquid absorption coefficients of this layer are evaluated (cldprop :199-262)
    int err = 0, icemode = -1, liqmode = -1, iceind = 0, liqind = 0, index_i = 1, index_l = 1;
    double radice = 0.0, ice_single = 0.0, fint_i = 0.0, liq_single = 0.0, fint_l = 0.0;
    if (enters && inflag == 2) {
        radice = (GCM ? g.reice : c.rei)[gi];
        if (ciwp == 0.0) { icemode = 0; ice_single = 0.0; iceind = 0; }
        else if (iceflag == 0) {
            if (radice < 10.0) err = E_ICE_SMALL;
            icemode = 0; ice_single = T.absice0[0] + T.absice0[1] / radice; iceind = 0;
        } else if (iceflag == 1) {
            if (radice < 13.0 || radice > 130.) err = E_ICE_BOUNDS;
            icemode = 1; iceind = 1;
        } else if (iceflag == 2) {
            if (radice < 5.0 || radice > 131.0) err = E_ICE_BOUNDS;
            const double factor = (radice - 2.) / 3.;
            index_i = (int)factor; if (index_i == 43) index_i = 42;
            fint_i = factor - (double)index_i;
            icemode = 2; iceind = 2;
        } else if (iceflag == 3) {
            if (radice < 5.0 || radice > 140.0) err = E_ICE_GEN_BOUNDS;
            const double factor = (radice - 2.) / 3.;
            index_i = (int)factor; if (index_i == 46) index_i = 45;
            fint_i = factor - (double)index_i;
            icemode = 3; iceind = 2;
        } else err = E_BAD_FLAG;
        if (clwp == 0.0) { liqmode = 0; liq_single = 0.0; liqind = 0; if (iceind == 1) iceind = 2; }
        else if (liqflag == 0) { liqmode = 0; liq_single = T.absliq0; liqind = 0; if (iceind == 1) iceind = 2; }
        else if (liqflag == 1) {
            const double radliq = (GCM ? g.reliq : c.rel)[gi];
            if (radliq < 2.5 || radliq > 60.) err = E_LIQ_BOUNDS;
            index_l = (int)(radliq - 1.5);
            if (index_l == 0) index_l = 1;
            if (index_l == 58) index_l = 57;
            fint_l = radliq - 1.5 - (double)index_l;
            liqmode = 1; liqind = 2;
        } else err = E_BAD_FLAG;
        if (err) index_i = clampi(index_i, 1, 42), index_l = clampi(index_l, 1, 57);
    }
    if (err) atomicCAS(W.err, 0, err);
    if (!cloudy) return;          // k_layer and the sweeps read odcld / efcl of cloudy layers only
    // cloud bands of the final count, each written to the spectral bands it serves
    for (int ib = 1; ib <= ncbands; ib++) {
        double t = 0.0;                                             // taucloud(lay, ib)
        if (enters && ib <= nb_at) {
            if (inflag == 0) t = GCM ? g.taucld[(ib - 1) + (size_t)NBND * gi] : c.tauc[gc + (size_t)nct * ((ib - 1) + (size_t)NBND * (lay - 1))];
            else if (inflag == 1) t = T.abscld1 * cwp;
            else if (inflag == 2) {
                int ki = icb_map(ib, iceind);
                const int kl = icb_map(ib, liqind);
                // iceflag = 1 (Ebert-Curry) has five ice bands.  When another layer of the column made cldprop carry 16 bands (liqflag = 1
                // with liquid water) and this layer is ice only, the reference promotes iceind to 2 and reads abscoice(6:16), values left over
                // from an earlier layer or call (src/rrtmg_lw_cldprop.f90:215-222,252-258,283-286: ill-defined upstream).  Bands 1-5 are the
                // reference's; bands 6-16 of such a layer use the fifth Ebert-Curry band here instead of reading past the 5 x 2 table.
                if (icemode == 1 && ki > 5) ki = 5;
                double ai, al;
                if (icemode == 0) ai = ice_single;
                else if (icemode == 1) ai = absice1[2 * (ki - 1)] + absice1[2 * (ki - 1) + 1] / radice;
                else if (icemode == 2) { const double *tb = absice2 + 43 * (ki - 1); ai = tb[index_i - 1] + fint_i * (tb[index_i] - tb[index_i - 1]); }
                else if (icemode == 3) { const double *tb = absice3 + 46 * (ki - 1); ai = tb[index_i - 1] + fint_i * (tb[index_i] - tb[index_i - 1]); }
                else ai = 0.0;
                if (liqmode == 0) al = liq_single;
                else if (liqmode == 1) { const double *tb = absliq1 + 58 * (kl - 1); al = tb[index_l - 1] + fint_l * (tb[index_l] - tb[index_l - 1]); }
                else al = 0.0;
                t = ciwp * ai + clwp * al;
            }
        }
        // optical depth along the diffusivity angle; secdiff is indexed by the CLOUD band ib (rtrn :323)
        const double od = W.percol[(size_t)(PC_SECDIFF + ib - 1) * ncb + col] * t;
        double ef = 0.0;
        if (mode == 1) ef = (1. - exp(-od)) * cf;
        // spectral bands served by cloud band ib: rtrn :343-349
        int Blo = ib, Bhi = ib;
        if (ncbands == 1) { Blo = 1; Bhi = NBND; }
        else if (ncbands == 5) { Blo = ib <= 2 ? ib : (ib == 3 ? 3 : (ib == 4 ? 6 : 9)); Bhi = ib <= 2 ? ib : (ib == 3 ? 5 : (ib == 4 ? 8 : NBND)); }
        for (int B = Blo; B <= Bhi; B++) {
            const size_t o = ((size_t)(B - 1) * nlay + (lay - 1)) * ncb + col;
            W.odcld[o] = od;
            if (mode == 1) W.efcl[o] = ef;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_blocksort : orders the 64-column blocks of a batch by their highest cloudy layer, deepest first (a counting sort, equal tops in
//               block order: the result does not depend on timing), pads the order to whole groups of SORT_GROUP with the block number
//               "one past the last" (columns past the end: every sweep lane of such a block is masked), and gives every group the
//               top of its first = deepest block as hand-off level.  The reference decides clear / cloudy per (column, layer)
//               (icldlyr, src/rrtmg_lw_rtrnmr.f90:509-704); this is that decision at the granularity of a wavefront's 64 columns:
//               above a block's hand-off level the total-sky stream IS the clear-sky stream.  One workgroup; nblk <= SORT_MAXBLK.
// ------------------------------------------------------------------------------------------------
constexpr int SORT_GROUP = 12;                  // divisible by every column-block count of a sweep workgroup (nsb_fit)
constexpr int SORT_MAXBLK = 16384;              // 1 048 576 columns per batch
constexpr int SORT_MAXLAY = 603;                // parrrtm.f90:31 mxlay
__host__ __device__ constexpr int nsb_fit(int n) { return n >= 12 ? 12 : n >= 6 ? 6 : n >= 4 ? 4 : n >= 3 ? 3 : n >= 2 ? 2 : 1; }
__host__ __device__ constexpr int sort_slots(int nblk) { return (nblk + SORT_GROUP - 1) / SORT_GROUP * SORT_GROUP; }

__global__ __launch_bounds__(256) void k_blocksort(Workspace W, int nblk, int force_top)
{
    __shared__ unsigned short s_top[SORT_MAXBLK];
    __shared__ int s_cnt[SORT_MAXLAY + 1], s_start[SORT_MAXLAY + 1], s_gtop[SORT_MAXBLK / SORT_GROUP + 1];
    const int tid = threadIdx.x, nth = blockDim.x, nlay = W.nlay, nslot = sort_slots(nblk), ngrp = nslot / SORT_GROUP;
    for (int v = tid; v <= nlay; v += nth) s_cnt[v] = 0;
    __syncthreads();
    for (int b = tid; b < nblk; b += nth) {
        const int t = min(max(W.btop[b], 0), nlay);
        s_top[b] = (unsigned short)t;
        atomicAdd(&s_cnt[t], 1);
    }
    __syncthreads();
    // first position of every value, deepest first: an exclusive prefix sum over u = nlay - v by the first wave (each lane a run of values)
    if (tid < 64) {
        const int per = (nlay + 64) / 64, u0 = tid * per;
        int sum = 0;
        for (int u = u0; u < u0 + per && u <= nlay; u++) sum += s_cnt[nlay - u];
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (tid >= o) incl += t; }
        int pos = incl - sum;
        for (int u = u0; u < u0 + per && u <= nlay; u++) { s_start[nlay - u] = pos; pos += s_cnt[nlay - u]; }
    }
    __syncthreads();
    // top of the first block of every group: the value whose run [start, start + count) holds position SORT_GROUP g (the starts grow with
    // u = nlay - v: binary search for the last u with start <= q; runs of count 0 share their start with the next non-empty one, which the
    // search passes over because it looks for the LAST such u ... of a non-empty run: step back over empty ones)
    for (int g = tid; g < ngrp; g += nth) {
        const int q = SORT_GROUP * g;           // (< nblk: the last group starts below nblk)
        int lo = 0, hi = nlay;                  // largest u with s_start[nlay - u] <= q
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_start[nlay - mid] <= q) lo = mid; else hi = mid - 1; }
        while (lo > 0 && s_cnt[nlay - lo] == 0) lo--;      // (an empty run at the end of equal starts)
        // force_top (a batch too small to fill the chip): every group hands off at the top of the column, i.e. the cloud-zone sweep walks
        // all levels - with its clear-sky body below the lowest cloud and its clear-sky branch above the highest - and the two clear-sky
        // launches, each a latency chain of its own, are not made (driver.hip: one_sweep).  Same numbers: a level's partial does not depend
        // on the kernel that swept it.
        const int top = force_top ? nlay : nlay - lo;
        s_gtop[g] = top;
        W.hgrp[g] = top;
    }
    __syncthreads();
    // placement by the first wave, 64 blocks at a time in block order: the lanes that hold the same value as the first unplaced lane take
    // consecutive positions of that value's run (a handful of distinct values per 64 blocks)
    if (tid < 64) {
        for (int b0 = 0; b0 < nblk; b0 += 64) {
            const int b = b0 + tid;
            const bool in = b < nblk;
            const int v = in ? (int)s_top[b] : -1;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(in);
            while (todo) {
                const int lead = __builtin_ctzll(todo);
                const int lv = __shfl(v, lead, 64);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(in && v == lv) & todo;
                const int base = s_start[lv];
                if (in && v == lv) {
                    const int p = base + __builtin_popcountll(m & ((1ull << tid) - 1ull));
                    W.order[p] = b;
                    W.hblk[b] = s_gtop[p / SORT_GROUP];
                }
                __builtin_amdgcn_wave_barrier();
                if (tid == lead) s_start[lv] = base + __builtin_popcountll(m);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                todo &= ~m;
            }
        }
    }
    for (int p = nblk + tid; p < nslot; p += nth) W.order[p] = nblk;
    __syncthreads();                    // (the order written by the first wave is read below)
    // lowest cloudy layer of every group (a group without cloud: nlay + 1; its hand-off level is 0 and nothing is swept in the zone)
    for (int g = tid; g < ngrp; g += nth) {
        int b = nlay + 1;
        for (int p = SORT_GROUP * g; p < SORT_GROUP * (g + 1) && p < nblk; p++) b = min(b, W.bbot[W.order[p]]);
        W.hbot[g] = b;
    }
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
#ifdef RRLW_G256
#define BAND_NG(NG_) 16
#else
#define BAND_NG(NG_) NG_
#endif
#define BAND_TRAITS(B_, NG_, LO, UP) \
    template <> struct BT<B_> { static constexpr int ng = BAND_NG(NG_); static constexpr Region lo = LO; static constexpr Region up = UP; };

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

__device__ const double kMult4[16] = {1, 1, 1, 1, 1, 1, 1, 0.92, 0.88, 1.07, 1.1, 0.99, 0.88, 0.943, 1, 1};   // taumol :1028-1034
__device__ const double kMult7[16] = {1, 1, 1, 1, 1, 0.92, 0.88, 1.07, 1.1, 0.99, 0.855, 1, 1, 1, 1, 1};      // taumol :1664-1669 (g 6-11 of the band, whatever its size)

// setcoef results of one (layer, column), kept in registers by k_layer
struct LayerCoef {
    double f[NFIELD];
    int jp, jt, jt1, indself, indfor, indminor;
};

struct Spec { double speccomb, specparm, fs; int js; };
__device__ __forceinline__ Spec spec_calc(double cola, double rat, double colb, double mult, double oneminus)
{
    Spec s;                                         // taumol :523-528
    s.speccomb = cola + rat * colb;
    s.specparm = fdiv(cola, s.speccomb);
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

// ------------------------------------------------------------------------------------------------
// taumol as a sparse linear combination.  For one (layer, column) and band, the gaseous optical depth of
// every g-point is   tau(g) = sum_i  w_i * K[off_i + g]   over a handful of table rows: the P/T(/mixture)
// interpolation stencil of the key species, the two self- and two foreign-continuum rows, 2 or 4 rows per
// minor gas, one row per halocarbon (src/rrtmg_lw_taumol.f90:299-3164).  rows_prep
// turns the layer's setcoef quantities into the (off_i, w_i) list once; rows_eval then needs only the g-offset.
// The weights are algebraically those of the reference (e.g. selffac*(1-selffrac), selffac*selffrac for
// selffac*(s0 + selffrac*(s1-s0))); products are associated differently, i.e. results differ at rounding level.
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int region_nrows(const Region &R, bool lower)
{
    int n = 0;
    if (R.key == K_SINGLE) n += 4;
    else if (R.key == K_BINARY) n += lower ? 12 : 8;
    if (R.self_) n += 2;
    if (R.for_) n += 2;
    for (int i = 0; i < R.nm; i++) n += R.m[i].two_d ? 4 : 2;
    n += R.ncfc;
    return n;
}

__host__ __device__ constexpr int region_major_rows(const Region &R, bool lower)
{
    return R.key == K_SINGLE ? 4 : (R.key == K_BINARY ? (lower ? 12 : 8) : 0);
}
__host__ __device__ constexpr int region_minor_base(const Region &R, bool lower, int im)
{
    int n = region_major_rows(R, lower) + (R.self_ ? 2 : 0) + (R.for_ ? 2 : 0);
    for (int i = 0; i < im; i++) n += R.m[i].two_d ? 4 : 2;
    return n;
}

// g-point "quads": every band is padded to a multiple of 4 g-points; quad q of the 38 holds g-points
// QG0(band) + 4*(q - QSTART(band)) ... +3 of one band.  Scratch arrays are [array][quad][layer][column][4].
#ifdef RRLW_G256
__host__ __device__ constexpr int band_ng(int B) { return B >= 1 ? 16 : 16; }
#else
__host__ __device__ constexpr int band_ng(int B) { return B == 1 ? 10 : B == 2 ? 12 : B == 3 ? 16 : B == 4 ? 14 : B == 5 ? 16 : B == 6 ? 8 : B == 7 ? 12 :
                                                    B == 8 ? 8 : B == 9 ? 12 : B == 10 ? 6 : B == 11 ? 8 : B == 12 ? 8 : B == 13 ? 4 : 2; }
#endif
__host__ __device__ constexpr int band_nquad(int B) { return (band_ng(B) + 3) / 4; }
__host__ __device__ constexpr int band_qstart(int B)     // not recursive: a recursive constexpr function that is not folded becomes a
{                                                       // real device call with a dynamic stack
    int s = 0;
    for (int b = 1; b < B; b++) s += band_nquad(b);
    return s;
}
constexpr int NQUAD = band_qstart(17);      // 38
__host__ __device__ constexpr int band_g0(int B)      // 0-based first g-point of band B among the 140
{
    int s = 0;
    for (int b = 1; b < B; b++) s += band_ng(b);
    return s;
}
static_assert(band_g0(17) == NGPT, "g-point table");
static_assert(NQUAD == (NGPT == 140 ? 38 : 64), "quad table");

// setcoef's chi_mls ratios and mixing ratios (6 x 59 and 7 x 59 doubles, contiguous in the static buffer): rows_prep reads two or three
// of them per band and needs them at once; k_layer copies them into LDS once per workgroup (a global read costs ~500 cycles of
// exposed latency each, 16 bands x 3).
constexpr int NRATCHI = 6 * 59 + 7 * 59;
__shared__ double s_ratchi[NRATCHI];

template <int N>
struct Rows {
    unsigned off[N > 0 ? N : 1];   // element offsets into the packed k-table buffer (row start, g = 0)
    double w[N > 0 ? N : 1];
    unsigned fw;                   // binary-key regions: Planck fractions are interpolated between rows js-1 and js of fracrefa/b with
                                   // weight fs (taumol :556-561, :692-693); packed (js << 28) | fs in 28-bit fixed point for the sweeps
};

// binary-key bands get a slot in Workspace::fw (every band whose upper region is binary-key has a binary-key lower region too)
template <int B> constexpr unsigned lo_binary_mask() { return (BT<B>::lo.key == K_BINARY ? 1u << (B - 1) : 0u) | lo_binary_mask<B - 1>(); }
template <> constexpr unsigned lo_binary_mask<0>() { return 0u; }
template <int B> constexpr unsigned up_binary_mask() { return (BT<B>::up.key == K_BINARY ? 1u << (B - 1) : 0u) | up_binary_mask<B - 1>(); }
template <> constexpr unsigned up_binary_mask<0>() { return 0u; }
template <int B> constexpr unsigned up_zero_mask() { return (BT<B>::up.key == K_ZERO ? 1u << (B - 1) : 0u) | up_zero_mask<B - 1>(); }
template <> constexpr unsigned up_zero_mask<0>() { return 0u; }
template <int B> constexpr unsigned up_from_a_mask() { return (BT<B>::up.frac_from_a ? 1u << (B - 1) : 0u) | up_from_a_mask<B - 1>(); }
template <> constexpr unsigned up_from_a_mask<0>() { return 0u; }
constexpr unsigned LO_BINARY = lo_binary_mask<16>(), UP_BINARY = up_binary_mask<16>(), UP_ZERO = up_zero_mask<16>(), UP_FROM_A = up_from_a_mask<16>();
static_assert((UP_BINARY & ~LO_BINARY) == 0u, "fw slots are assigned by the lower region");
__host__ __device__ constexpr int popcnt_c(unsigned v) { int n = 0; for (; v; v &= v - 1) n++; return n; }
constexpr int NFW = popcnt_c(LO_BINARY);                                          // 9
__host__ __device__ constexpr int fw_slot(int B) { return popcnt_c(LO_BINARY & ((1u << (B - 1)) - 1u)); }

// one minor gas: rows IBASE.. of the list
template <int B, bool LOWER, int N, int IM>
__device__ __forceinline__ void rows_prep_minor(const DevTables &T, const LayerCoef &C, Rows<N> &rw)
{
    constexpr Region R = LOWER ? BT<B>::lo : BT<B>::up;
    constexpr unsigned ng = BT<B>::ng;
    constexpr Minor M = R.m[IM];
    constexpr int I0 = region_minor_base(R, LOWER, IM);
    const BandLayout &L = T.band[B - 1];
    const double mf = C.f[F_MINORFRAC];
    double amount;
    if constexpr (M.amt == A_COL) amount = C.f[F_COLH2O + M.sp];
    else if constexpr (M.amt == A_BRD_N2) amount = C.f[F_COLBRD] * C.f[F_SCALEMINORN2];
    else if constexpr (M.amt == A_O2) amount = C.f[F_COLO2] * C.f[F_SCALEMINOR];
    else if constexpr (M.amt == A_BRD) amount = C.f[F_COLBRD] * C.f[F_SCALEMINOR];
    else {                                              // A_ADJ: taumol :547-554
        const double colx = C.f[F_COLH2O + M.sp], coldry = C.f[F_COLDRY];
        const double chiref = M.chiconst > 0. ? M.chiconst : s_ratchi[(T.sl.chi - T.sl.rat) + M.sp * 59 + C.jp];     // chi_mls(sp+1, jp+1)
        const double chi = fdiv(colx, coldry);
        const double ratx = fdiv(1.e20 * chi, chiref);
        amount = colx;
        if (ratx > M.thr) amount = (M.base + pow(ratx - M.base, M.expo)) * chiref * coldry * 1.e-20;
    }
    const unsigned slot = (unsigned)(LOWER ? L.minor_lo[IM] : L.minor_up[IM]);
    if constexpr (M.two_d) {                            // taumol :635-639
        constexpr unsigned nj = LOWER ? 9 : 5;
        const Spec sm = spec_calc(C.f[F_COLH2O + R.a], T.refrat[B - 1][M.refslot], C.f[F_COLH2O + R.b], LOWER ? 8. : 4., T.oneminus);
        const unsigned r = slot + ((unsigned)(C.indminor - 1) * nj + (unsigned)(sm.js - 1)) * ng;
        rw.off[I0] = r;                    rw.w[I0] = amount * ((1. - mf) * (1. - sm.fs));
        rw.off[I0 + 1] = r + ng;           rw.w[I0 + 1] = amount * ((1. - mf) * sm.fs);
        rw.off[I0 + 2] = r + nj * ng;      rw.w[I0 + 2] = amount * (mf * (1. - sm.fs));
        rw.off[I0 + 3] = r + nj * ng + ng; rw.w[I0 + 3] = amount * (mf * sm.fs);
    } else {
        const unsigned r = slot + (unsigned)(C.indminor - 1) * ng;
        rw.off[I0] = r;          rw.w[I0] = amount * (1. - mf);
        rw.off[I0 + 1] = r + ng; rw.w[I0 + 1] = amount * mf;
    }
}

template <int B, bool LOWER, int N>
__device__ __forceinline__ void rows_prep(const DevTables &T, const LayerCoef &C, Rows<N> &rw)
{
    constexpr Region R = LOWER ? BT<B>::lo : BT<B>::up;
    constexpr unsigned ng = BT<B>::ng;
    const BandLayout &L = T.band[B - 1];
    const int jp = C.jp, jt = C.jt, jt1 = C.jt1;
    const double *rat_tab = s_ratchi;
    rw.fw = 0u;
    if constexpr (R.key == K_ZERO) return;

    double corradj = 1.;
    if constexpr (R.corr == C_B1LO) { if (C.f[F_PAVEL] < 250.) corradj = 1. - 0.15 * (250. - C.f[F_PAVEL]) / 154.4; }
    else if constexpr (R.corr == C_B1UP) corradj = 1. - 0.15 * (C.f[F_PAVEL] / 95.6);
    else if constexpr (R.corr == C_B2LO) corradj = 1. - .05 * (C.f[F_PAVEL] - 100.) / 900.;

    if constexpr (R.key == K_SINGLE) {                      // taumol :356-360
        const unsigned r0 = LOWER ? ((jp - 1) * 5 + (jt - 1)) : ((jp - 13) * 5 + (jt - 1));
        const unsigned r1 = LOWER ? (jp * 5 + (jt1 - 1)) : ((jp - 12) * 5 + (jt1 - 1));
        const unsigned tab = (unsigned)(LOWER ? L.absa : L.absb);
        const double colk = C.f[F_COLH2O + R.a];
        rw.off[0] = tab + r0 * ng;       rw.w[0] = colk * C.f[F_FAC00];
        rw.off[1] = tab + (r0 + 1) * ng; rw.w[1] = colk * C.f[F_FAC10];
        rw.off[2] = tab + r1 * ng;       rw.w[2] = colk * C.f[F_FAC01];
        rw.off[3] = tab + (r1 + 1) * ng; rw.w[3] = colk * C.f[F_FAC11];
    } else if constexpr (R.key == K_BINARY) {
        const double cola = C.f[F_COLH2O + R.a], colb = C.f[F_COLH2O + R.b];
        const double rat = rat_tab[R.rat * 59 + (jp - 1)], rat_1 = rat_tab[R.rat * 59 + jp];
        constexpr double mult = LOWER ? 8. : 4.;
        const Spec s = spec_calc(cola, rat, colb, mult, T.oneminus);
        const Spec s1 = spec_calc(cola, rat_1, colb, mult, T.oneminus);
        if constexpr (LOWER) {                              // taumol :563-663
            const unsigned tab = (unsigned)L.absa;
            const int ind0 = ((jp - 1) * 5 + (jt - 1)) * 9 + s.js - 1;      // 0-based row
            const int ind1 = (jp * 5 + (jt1 - 1)) * 9 + s1.js - 1;
            int so;
            double w6[6];
            stencil6(s.specparm, s.fs, C.f[F_FAC00], C.f[F_FAC10], so, w6);
            unsigned r = tab + (unsigned)(ind0 + so) * ng;
            rw.off[0] = r;           rw.w[0] = s.speccomb * w6[0];
            rw.off[1] = r + ng;      rw.w[1] = s.speccomb * w6[1];
            rw.off[2] = r + 2 * ng;  rw.w[2] = s.speccomb * w6[2];
            rw.off[3] = r + 9 * ng;  rw.w[3] = s.speccomb * w6[3];
            rw.off[4] = r + 10 * ng; rw.w[4] = s.speccomb * w6[4];
            rw.off[5] = r + 11 * ng; rw.w[5] = s.speccomb * w6[5];
            stencil6(s1.specparm, s1.fs, C.f[F_FAC01], C.f[F_FAC11], so, w6);
            r = tab + (unsigned)(ind1 + so) * ng;
            rw.off[6] = r;            rw.w[6] = s1.speccomb * w6[0];
            rw.off[7] = r + ng;       rw.w[7] = s1.speccomb * w6[1];
            rw.off[8] = r + 2 * ng;   rw.w[8] = s1.speccomb * w6[2];
            rw.off[9] = r + 9 * ng;   rw.w[9] = s1.speccomb * w6[3];
            rw.off[10] = r + 10 * ng; rw.w[10] = s1.speccomb * w6[4];
            rw.off[11] = r + 11 * ng; rw.w[11] = s1.speccomb * w6[5];
        } else {                                            // taumol :749-771
            const unsigned tab = (unsigned)L.absb;
            const unsigned ind0 = ((jp - 13) * 5 + (jt - 1)) * 5 + s.js - 1;
            const unsigned ind1 = ((jp - 12) * 5 + (jt1 - 1)) * 5 + s1.js - 1;
            unsigned r = tab + ind0 * ng;
            rw.off[0] = r;          rw.w[0] = s.speccomb * ((1. - s.fs) * C.f[F_FAC00]);
            rw.off[1] = r + ng;     rw.w[1] = s.speccomb * (s.fs * C.f[F_FAC00]);
            rw.off[2] = r + 5 * ng; rw.w[2] = s.speccomb * ((1. - s.fs) * C.f[F_FAC10]);
            rw.off[3] = r + 6 * ng; rw.w[3] = s.speccomb * (s.fs * C.f[F_FAC10]);
            r = tab + ind1 * ng;
            rw.off[4] = r;          rw.w[4] = s1.speccomb * ((1. - s1.fs) * C.f[F_FAC01]);
            rw.off[5] = r + ng;     rw.w[5] = s1.speccomb * (s1.fs * C.f[F_FAC01]);
            rw.off[6] = r + 5 * ng; rw.w[6] = s1.speccomb * ((1. - s1.fs) * C.f[F_FAC11]);
            rw.off[7] = r + 6 * ng; rw.w[7] = s1.speccomb * (s1.fs * C.f[F_FAC11]);
        }
        // Planck fractions interpolated in the mixture: taumol :556-561, :692-693
        const Spec sp = spec_calc(cola, T.refrat[B - 1][R.planck_slot], colb, mult, T.oneminus);
        unsigned q = (unsigned)(sp.fs * 268435456.0 + 0.5);
        if (q > 268435455u) q = 268435455u;
        rw.fw = ((unsigned)sp.js << 28) | q;
    }
    constexpr int IS = region_major_rows(R, LOWER);
    if constexpr (R.self_) {                                // taumol :350-351
        const unsigned r = (unsigned)L.selfref + (unsigned)(C.indself - 1) * ng;
        rw.off[IS] = r;          rw.w[IS] = C.f[F_SELFFAC] * (1. - C.f[F_SELFFRAC]);
        rw.off[IS + 1] = r + ng; rw.w[IS + 1] = C.f[F_SELFFAC] * C.f[F_SELFFRAC];
    }
    constexpr int IF = IS + (R.self_ ? 2 : 0);
    if constexpr (R.for_) {                                 // taumol :352-353
        const unsigned r = (unsigned)L.forref + (unsigned)(C.indfor - 1) * ng;
        rw.off[IF] = r;          rw.w[IF] = C.f[F_FORFAC] * (1. - C.f[F_FORFRAC]);
        rw.off[IF + 1] = r + ng; rw.w[IF + 1] = C.f[F_FORFAC] * C.f[F_FORFRAC];
    }
    if constexpr (R.nm > 0) rows_prep_minor<B, LOWER, N, 0>(T, C, rw);
    if constexpr (R.nm > 1) rows_prep_minor<B, LOWER, N, 1>(T, C, rw);
    if constexpr (R.nm > 2) rows_prep_minor<B, LOWER, N, 2>(T, C, rw);
    constexpr int IC = region_minor_base(R, LOWER, R.nm);   // halocarbons: taumol :1254, :1381-1382, :1753-1754
    if constexpr (R.ncfc > 0) { rw.off[IC] = (unsigned)L.vec[0]; rw.w[IC] = C.f[F_WX1 + R.cfc_wx[0] - 1]; }
    if constexpr (R.ncfc > 1) { rw.off[IC + 1] = (unsigned)L.vec[1]; rw.w[IC + 1] = C.f[F_WX1 + R.cfc_wx[1] - 1]; }
    static_assert(IC + R.ncfc == N, "row count");
    if constexpr (R.corr != C_NONE) {
#pragma unroll
        for (int i = 0; i < N; i++) rw.w[i] = corradj * rw.w[i];
    }
}

// 16-byte table load through a buffer descriptor: 32-bit per-lane byte offset, no 64-bit address arithmetic,
// and the hardware range check returns zeros for anything outside the table buffer.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 ld2(__amdgpu_buffer_rsrc_t rsrc, unsigned elem_off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(elem_off * 8u), 0, 0);
    double2 d;
    __builtin_memcpy(&d, &v, 16);
    return d;
}

// the same 16 bytes from the band's tables staged in LDS (element offset into the staging buffer)
__device__ __forceinline__ double2 ldl(const double2 *lds, unsigned elem_off) { return lds[elem_off >> 1]; }

// Workgroup barrier that orders LDS accesses only.  __syncthreads() is a workgroup-scope release / acquire over ALL address spaces: on
// gfx950 that puts `s_waitcnt vmcnt(0)` in front of the barrier, and k_layer's waves then sit out the acknowledgement of every code
// store of the band they have just finished (stores count in vmcnt) sixteen times per layer although no other wave reads those codes.
__device__ __forceinline__ void lds_barrier()
{
#if defined(RRLW_LAYER_SYNCTHREADS)
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

// In-kernel stamps (diagnostic build -DRRLW_LAYER_STAMPS only: where a k_layer wave spends its cycles; the build's run time means
// nothing, its SHARES do).  Segment sums per wave in LDS, added to g_stamps at the end of the kernel; rrtmg_lw_hip_debug_stamps reads them.
#ifdef RRLW_LAYER_STAMPS
constexpr int NSTAMP = 8;
__device__ unsigned long long g_stamps[NSTAMP + 1];
__shared__ unsigned long long s_stamp[16 * (NSTAMP + 1)];
__device__ __forceinline__ void stamp(int seg)
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *w = s_stamp + (threadIdx.x >> 6) * (NSTAMP + 1);
        if (seg >= 0) w[seg] += t - w[NSTAMP];
        w[NSTAMP] = t;
    }
}
#define STAMP(seg) stamp(seg)
#else
#define STAMP(seg) ((void)0)
#endif

// ------------------------------------------------------------------------------------------------
// Staging of the bands' tables in LDS.  The table evaluation returns ~15 KB of table rows per (layer, column) to registers; through the
// vector L1 that is bound by its 64 B/clk return path (measured: 90 % of it), LDS returns 256 B/clk.  All threads of a k_layer
// workgroup work on the SAME layer, so they need the same few pressure planes of the key-species tables and the same few temperature
// slices of the minor-gas tables: the workgroup stages, per band, the planes jp0 .. jp0+2 (jp0 = its smallest reference-pressure index;
// a cell uses planes jp and jp+1), MINOR_WIN slices of each minor-gas table from im0 = its smallest minor-gas temperature index (a
// cell uses slices indminor and indminor+1), the whole self / foreign / halocarbon tables, and evaluates from LDS.  A wave with a cell
// outside the window (other region of the atmosphere, jp > jp0 + 1, indminor > im0 + MINOR_WIN - 2) evaluates from global memory -
// same values either way.
// Bands are staged in PASSES of several bands (LayerPasses, below): in-kernel stamps (tools/stamps_run.py) showed a wave spending 30 %
// of its cycles between the barriers of the sixteen per-band staging rounds and 7 % at the barrier in front of each; a pass is ONE
// round - the tables of up to eleven bands, as one flat list of 16-byte pieces - and the windowed minor-gas tables make four passes
// of the sixteen bands.
// ------------------------------------------------------------------------------------------------
enum Role { RL_MAJOR, RL_SELF, RL_FOR, RL_MINOR0, RL_MINOR1, RL_MINOR2, RL_CFC0, RL_CFC1, NROLE };

#ifndef RRLW_MINOR_WIN
#define RRLW_MINOR_WIN 6          // temperature slices (7.2 K each) of a minor-gas table staged per workgroup; 19 = the whole table
#endif
constexpr int MINOR_WIN = RRLW_MINOR_WIN;
static_assert(MINOR_WIN >= 2 && MINOR_WIN <= 19, "minor-gas tables have 19 temperature slices");
// The staging window of a workgroup comes in two sizes (round 5, profiles/round5_orography.md).  NARROW - three pressure planes (cells
// with jp0 <= jp <= jp0 + 1), MINOR_WIN minor-gas slices - where the 256 columns of the layer lie that close together: a pressure grid
// that is (nearly) the same in every column.  WIDE - five planes (jp0 .. jp0 + 3), ten slices, the bands in more passes - where they do
// not: a terrain-following grid puts the columns of one model level at surface-pressure factors of 0.55 .. 1.03 (three reference planes
// apart, setcoef :276-284) and at temperatures 30 K apart.  Chosen per workgroup from the spread of its cells; with the narrow window
// alone half of such a grid's waves fell to the global-memory evaluation and k_layer took 37.7 ms instead of 23 per 1e6 columns.
template <int NPL_, int MW_> struct StageWin { static constexpr int NPL = NPL_, MW = MW_; };
using WinNarrow = StageWin<3, MINOR_WIN>;
#ifdef RRLW_G256
using WinWide = WinNarrow;               // (the 256-g-point build keeps the narrow window: its tables are twice the size)
constexpr bool HAVE_WIDE = false;
#else
using WinWide = StageWin<5, 10>;
constexpr bool HAVE_WIDE = true;
#endif

template <int B, bool LOWER, class WN>
struct Stage {
    static constexpr Region R = LOWER ? BT<B>::lo : BT<B>::up;
    static constexpr int ng = BT<B>::ng;
    static constexpr int nsp = R.key == K_BINARY ? (LOWER ? 9 : 5) : 1;          // mixture points per (P, T) node
    static constexpr int NPL = WN::NPL;                                          // pressure planes staged
    static constexpr bool has_major = R.key == K_SINGLE || R.key == K_BINARY;
    static constexpr int major_rows = has_major ? NPL * 5 * nsp + 3 : 0;         // + 3: zero-weight stencil rows may lie past the last plane
    static constexpr int nj(int k) { return R.m[k].two_d ? (LOWER ? 9 : 5) : 1; }   // rows per temperature slice of minor gas k
    static constexpr int mrows(int k) { return k < R.nm ? WN::MW * nj(k) : 0; }
    // doubles of the segment with role `r`, and its start within the band's part of the staging buffer (segments follow each other)
    static constexpr int n(int r)
    {
        return r == RL_MAJOR ? major_rows * ng : r == RL_SELF ? (R.self_ ? 10 * ng : 0) : r == RL_FOR ? (R.for_ ? 4 * ng : 0) :
               r == RL_MINOR0 ? mrows(0) * ng : r == RL_MINOR1 ? mrows(1) * ng : r == RL_MINOR2 ? mrows(2) * ng :
               r == RL_CFC0 ? (R.ncfc > 0 ? ng : 0) : (R.ncfc > 1 ? ng : 0);
    }
    static constexpr int l(int r) { int s = 0; for (int q = 0; q < r; q++) s += n(q); return s; }
    static constexpr int total = l(NROLE);
};

#ifndef RRLW_LAYER_BLOCK
#define RRLW_LAYER_BLOCK 256      // (measured: 192 threads 29.9 ms, 256 25.5, 384 43.0, 512 35.5 per 1e6 cloudy columns)
#endif
constexpr int LAYER_BLOCK = RRLW_LAYER_BLOCK;   // threads of a k_layer workgroup: three workgroups per CU (a 47 KB staging buffer each), three waves per SIMD
#ifndef RRLW_STAGE_DOUBLES
#define RRLW_STAGE_DOUBLES 5800   // capacity of the staging buffer: three workgroups per CU, each 46 400 B + 6 136 B of chi / ratio tables + flags = 52 552 B (the LDS is handed out in pieces of 2 KB: 53 248 x 3 of the 163 840 B)
#endif
constexpr int STAGE_DOUBLES = RRLW_STAGE_DOUBLES;

// the bands of one staging pass, in evaluation order
template <int... Bs> struct BandList { static constexpr int b[sizeof...(Bs)] = {Bs...}; static constexpr int n = sizeof...(Bs); };
// start of band number I of the pass within the staging buffer, in doubles
template <class PL, bool LOWER, class WN, int I> constexpr int pass_base()
{
    if constexpr (I == 0) return 0;
    else return pass_base<PL, LOWER, WN, I - 1>() + Stage<PL::b[I - 1], LOWER, WN>::total;
}
template <class PL, bool LOWER, class WN> constexpr int pass_total() { return pass_base<PL, LOWER, WN, PL::n>(); }

// where each segment of band B comes from: element offsets into the packed table buffer (wave-uniform: scalar registers)
template <int B, bool LOWER, class WN>
__device__ __forceinline__ void stage_sources(const DevTables &T, int jp0, int im0, unsigned (&g)[NROLE])
{
    using S = Stage<B, LOWER, WN>;
    constexpr Region R = S::R;
    constexpr int ng = S::ng;
    const BandLayout &L = T.band[B - 1];
#pragma unroll
    for (int r = 0; r < NROLE; r++) g[r] = 0u;
    if constexpr (S::has_major) {
        const int plane0 = LOWER ? jp0 - 1 : jp0 - 13;
        g[RL_MAJOR] = (unsigned)(LOWER ? L.absa : L.absb) + (unsigned)(plane0 * 5 * S::nsp * ng);
    }
    if constexpr (R.self_) g[RL_SELF] = (unsigned)L.selfref;
    if constexpr (R.for_) g[RL_FOR] = (unsigned)L.forref;
    if constexpr (R.nm > 0) g[RL_MINOR0] = (unsigned)(LOWER ? L.minor_lo[0] : L.minor_up[0]) + (unsigned)((im0 - 1) * S::nj(0) * ng);
    if constexpr (R.nm > 1) g[RL_MINOR1] = (unsigned)(LOWER ? L.minor_lo[1] : L.minor_up[1]) + (unsigned)((im0 - 1) * S::nj(1) * ng);
    if constexpr (R.nm > 2) g[RL_MINOR2] = (unsigned)L.minor_lo[2] + (unsigned)((im0 - 1) * S::nj(2) * ng);
    if constexpr (R.ncfc > 0) g[RL_CFC0] = (unsigned)L.vec[0];
    if constexpr (R.ncfc > 1) g[RL_CFC1] = (unsigned)L.vec[1];
}

// The staging buffer of a pass is the bands' segments one after the other; a thread copies the 16-byte pieces tid, tid + 256, ...
// Per-lane source offsets of those pieces: of the segments, whose bounds are compile-time constants, only the few that meet the 256
// pieces of round k contribute a compare and a select.
template <class PL, bool LOWER, class WN, int I, int IT>
__device__ __forceinline__ void pass_offsets(const DevTables &T, int jp0, int im0, int tid, unsigned (&off)[IT])
{
    if constexpr (I < PL::n) {
        using S = Stage<PL::b[I], LOWER, WN>;
        constexpr int base = pass_base<PL, LOWER, WN, I>();
        unsigned g[NROLE];
        stage_sources<PL::b[I], LOWER, WN>(T, jp0, im0, g);
#pragma unroll
        for (int r = 0; r < NROLE; r++) {
            const int n = S::n(r), lo = (base + S::l(r)) / 2, hi = lo + n / 2;         // constants once the loops are unrolled
            if (n > 0) {
#pragma unroll
                for (int k = 0; k < IT; k++) {
                    if (hi > LAYER_BLOCK * k && lo < LAYER_BLOCK * (k + 1)) {
                        const int c = tid + LAYER_BLOCK * k;
                        const unsigned src = g[r] + 2u * (unsigned)(c - lo);
                        off[k] = (lo <= LAYER_BLOCK * k || c >= lo) ? src : off[k];
                    }
                }
            }
        }
        pass_offsets<PL, LOWER, WN, I + 1, IT>(T, jp0, im0, tid, off);
    }
}

// two-phase copy: every thread first issues ALL its 16-byte loads (at most 12, independent), then writes them to LDS - a pass costs one
// memory round trip
template <class PL, bool LOWER, class WN>
__device__ __forceinline__ void stage_pass(const DevTables &T, __amdgpu_buffer_rsrc_t kt, double2 *lds, int jp0, int im0, int tid)
{
    constexpr int NCH = pass_total<PL, LOWER, WN>() / 2;
    constexpr int IT = (NCH + LAYER_BLOCK - 1) / LAYER_BLOCK;
    static_assert(pass_total<PL, LOWER, WN>() <= STAGE_DOUBLES, "the pass does not fit the staging buffer");
    if constexpr (IT > 0) {
        unsigned off[IT];
#pragma unroll
        for (int k = 0; k < IT; k++) off[k] = 0u;
        pass_offsets<PL, LOWER, WN, 0, IT>(T, jp0, im0, tid, off);
        double2 v[IT];
#pragma unroll
        for (int k = 0; k < IT; k++) v[k] = ld2(kt, off[k]);          // (past the last piece: some table data or, past the buffer, zeros; not stored)
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const int c = tid + LAYER_BLOCK * k;
            if (c < NCH) lds[c] = v[k];
        }
    }
}

// packed-table offset -> staging-buffer offset, per role, for band B whose part of the buffer starts at BASE
template <int B, bool LOWER, class WN, int BASE>
__device__ __forceinline__ void band_delta(const DevTables &T, int jp0, int im0, unsigned (&delta)[NROLE])
{
    unsigned g[NROLE];
    stage_sources<B, LOWER, WN>(T, jp0, im0, g);
#pragma unroll
    for (int r = 0; r < NROLE; r++) delta[r] = (unsigned)(BASE + Stage<B, LOWER, WN>::l(r)) - g[r];
}

// row offsets of the packed table buffer -> offsets into the staging buffer (rows_prep's row order: key species, self, foreign, minors, halocarbons)
template <int B, bool LOWER, int N>
__device__ __forceinline__ void rows_to_lds(Rows<N> &rw, const unsigned (&delta)[NROLE])
{
    constexpr Region R = LOWER ? BT<B>::lo : BT<B>::up;
    constexpr int IS = region_major_rows(R, LOWER), IF = IS + (R.self_ ? 2 : 0), IM0 = IF + (R.for_ ? 2 : 0);
    constexpr int IM1 = region_minor_base(R, LOWER, 1), IM2 = region_minor_base(R, LOWER, 2), IC = region_minor_base(R, LOWER, R.nm);
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int role = i < IS ? RL_MAJOR : i < IF ? RL_SELF : i < IM0 ? RL_FOR :
                         (i < IC ? (R.nm > 1 && i >= IM1 ? (R.nm > 2 && i >= IM2 ? RL_MINOR2 : RL_MINOR1) : RL_MINOR0) : (i == IC ? RL_CFC0 : RL_CFC1));
        rw.off[i] = rw.off[i] + delta[role];
#ifdef RRLW_KO_LDS_UNIFORM      // knock-out (timing only, wrong results): every lane reads the first lane's rows - pure broadcasts, no bank conflict
        rw.off[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)rw.off[i]);
#endif
    }
}

// Optical depth of ALL g-points of band B (padded to whole quads with zeros).  The table loads of a band are all independent, so
// they are issued as a software pipeline: the band's loads form one list (row-major: N rows x ng/2 16-byte loads), cut into chunks
// of RRLW_LOAD_CHUNK; chunk c+1 is in flight while chunk c is consumed.  sched_barrier keeps the compiler from hoisting every load
// to the top (which spills) or sinking them to their uses (which serialises the latency again).  LDS = the rows come from the
// workgroup's staging buffer (rw.off then holds staging offsets, rows_to_lds), otherwise from global memory through the vector L1.
#ifndef RRLW_LOAD_CHUNK
#define RRLW_LOAD_CHUNK 2       // loads in flight per pipeline stage.  From LDS two suffice (8 were needed through the vector L1) and the
#endif                          // registers saved allow three waves per SIMD: 36.4 vs 40.9 ms per 1e6 cloudy columns
#ifndef RRLW_CLOUD_QUADS
#define RRLW_CLOUD_QUADS 1      // quads of a band whose cloudy-layer look-ups are in flight together (2 until round 5: with the quad-wise table indices two spill ten registers)
#endif

template <int B, bool LOWER, int N>
struct BandLoads {
    static constexpr Region R = LOWER ? BT<B>::lo : BT<B>::up;
    static constexpr int ng = BT<B>::ng;
#ifdef RRLW_KO_HALFG       // knock-out (timing only, wrong results): only the first half of a band's g-points is evaluated - the occupancy experiment of round 3
    static constexpr int HP = (ng / 2 + 1) / 2;
#else
    static constexpr int HP = ng / 2;                                    // 16-byte loads per table row
#endif
    static constexpr int NK = N * HP;                                    // absorption-coefficient loads
    static constexpr int NL = NK;
    static constexpr int CH = RRLW_LOAD_CHUNK;
    static constexpr int NCH = (NL + CH - 1) / CH;

    template <int C, bool LDS>
    static __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t kt, const double2 *lds, const Rows<N> &rw, double2 (&b)[CH])
    {
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int idx = C * CH + k;
            if (idx < NK) b[k] = LDS ? ldl(lds, rw.off[idx / HP] + 2u * (unsigned)(idx % HP)) : ld2(kt, rw.off[idx / HP] + 2u * (unsigned)(idx % HP));
        }
    }

    template <int C>
    static __device__ __forceinline__ void consume(const Rows<N> &rw, const double2 (&b)[CH], double *tau)
    {
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int idx = C * CH + k;
            if (idx < NK) {
                const int i = idx / HP, p = idx % HP;
                tau[2 * p] = tau[2 * p] + rw.w[i] * b[k].x;
                tau[2 * p + 1] = tau[2 * p + 1] + rw.w[i] * b[k].y;
            }
        }
    }

    template <int C, bool LDS>
    static __device__ __forceinline__ void step(__amdgpu_buffer_rsrc_t kt, const double2 *lds, const Rows<N> &rw, double2 (&cur)[CH], double2 (&nxt)[CH],
                                                double *tau)
    {
        if constexpr (C < NCH) {
            if constexpr (C + 1 < NCH) issue<C + 1, LDS>(kt, lds, rw, nxt);
            __builtin_amdgcn_sched_barrier(0);
            consume<C>(rw, cur, tau);
            __builtin_amdgcn_sched_barrier(0);
            step<C + 1, LDS>(kt, lds, rw, nxt, cur, tau);
        }
    }
};

// tau0: what the sum starts from - the layer's aerosol optical depth of the band (taut = taug + taua, src/rrtmg_lw_rad.nomcica.f90:527-539),
// so that the sum arrives complete (one addition per cell less; with taua = 0, the usual case, bit for bit what it was).  The two regions
// whose gas optical depth is scaled per g-point afterwards (bands 4 and 7 above the tropopause) start from 0 and add taua behind the factor.
template <int B, bool LOWER, int N, bool LDS>
__device__ __forceinline__ void rows_eval_band(__amdgpu_buffer_rsrc_t kt, const double2 *lds, const Rows<N> &rw, double tau0, double (&tau)[4 * band_nquad(B)])
{
    using BL = BandLoads<B, LOWER, N>;
    constexpr int NP = 4 * band_nquad(B);
    constexpr bool SCALED = BL::R.mult != 0;
#pragma unroll
    for (int j = 0; j < NP; j++) tau[j] = (SCALED || j >= BL::ng) ? 0.0 : tau0;
    if constexpr (BL::NL > 0) {
        double2 b0[BL::CH], b1[BL::CH];
        BL::template issue<0, LDS>(kt, lds, rw, b0);
        BL::template step<0, LDS>(kt, lds, rw, b0, b1, tau);
    }
    if constexpr (SCALED) {
#pragma unroll
        for (int j = 0; j < BL::ng; j++) {
            if constexpr (BL::R.mult == 4) tau[j] = fma(tau[j], kMult4[j], tau0);
            else if constexpr (BL::R.mult == 7) tau[j] = fma(tau[j], kMult7[j], tau0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_layer : for one (column, layer):  inatm's layer part (src/rrtmg_lw_rad.nomcica.f90:785-867), setcoef
//           (src/rrtmg_lw_setcoef.f90:276-429), taumol for all bands (src/rrtmg_lw_taumol.f90:299-3164), taut = taug +
//           taua (src/rrtmg_lw_rad.nomcica.f90:527-539) and, per g-point, the optical depth along the diffusivity angle and
//           the branch rtrn's sweep body takes on it (src/rrtmg_lw_rtrn.f90:362-451) as a 4-byte code (cell_code).
//           Results go to the [quad][layer][column][4] code arrays.
// ------------------------------------------------------------------------------------------------
struct LayerArgs {
    int ncol, col0, nct, idrv, istart, iend;
    // a batch that does not fill the chip: the bands of a (window, layer) over `nparts` workgroups (grid.z, or grid.y of the wide launch),
    // workgroup p takes the bands of partmask[p] (bit B - 1); 0 / 1: one workgroup takes them all
    int nparts;
    unsigned partmask[4];
    int ktab_bytes;            // size of the packed k-table buffer (buffer descriptor range)
    const double *tauaer;      // (nct,nlay,16)
};

struct alignas(16) scr4 { scr_t v[4]; };

// The per-cell codes are written once by k_layer and read twice (down and up sweep) by k_sweepc / k_sweepz, gigabytes later: streaming
// (non-temporal) stores here and loads there (bload_scr4_nt) keep them from evicting the absorption tables from L2.
typedef float scr_vec __attribute__((ext_vector_type(4)));
// a quad's four codes as they lie in memory
struct pk4 { unsigned w[CODE_WORDS]; };
// Reduced code widths (measurement variants): bit (CODE_BITS - 1) = table cell, then the table index (<= 10000); otherwise the series-branch
// optical depth (0 <= od <= 0.06) as a small float - its float32 pattern scaled so that the exponent field fits E bits, cut to M mantissa
// bits with rounding; tiny values go through the scaled float's denormals, i.e. they keep an ABSOLUTE resolution.
//   24 bits: E = 5, M = 18 (relative 1.9e-6), scale 2^-91;   16 bits: E = 4, M = 11 (relative 2.4e-4), scale 2^-107
template <int BITS> struct CodeFmt;
template <> struct CodeFmt<24> { static constexpr int SH = 5;  static constexpr unsigned FLAG = 0x800000u, MASK = 0x7fffffu; static constexpr float DN = 0x1p-91f,  UP = 0x1p+91f; };
template <> struct CodeFmt<16> { static constexpr int SH = 12; static constexpr unsigned FLAG = 0x8000u,   MASK = 0x7fffu;   static constexpr float DN = 0x1p-107f, UP = 0x1p+107f; };
template <int BITS> __device__ __forceinline__ unsigned code_narrow(scr_t c)
{
    using F = CodeFmt<BITS>;
    if (c < 0.f) return F::FLAG | (unsigned)(int)(-c);
    const unsigned u = __float_as_uint(c * F::DN) & 0x7fffffffu;
    return min((u + (1u << (F::SH - 1))) >> F::SH, F::MASK);
}
template <int BITS> __device__ __forceinline__ scr_t code_widen(unsigned h)
{
    using F = CodeFmt<BITS>;
    const float tbl = -(float)(h & 0x3fffu), ser = __uint_as_float((h & F::MASK) << F::SH) * F::UP;
    return (h & F::FLAG) ? tbl : ser;
}
__device__ __forceinline__ pk4 pack4(const scr4 &v)
{
    pk4 p;
    if constexpr (CODE_BITS == 32) {
#pragma unroll
        for (int k = 0; k < 4; k++) p.w[k] = __float_as_uint(v.v[k]);
    } else if constexpr (CODE_BITS == 16) {
        p.w[0] = code_narrow<16>(v.v[0]) | (code_narrow<16>(v.v[1]) << 16);
        p.w[1] = code_narrow<16>(v.v[2]) | (code_narrow<16>(v.v[3]) << 16);
    } else {
        const unsigned a = code_narrow<24>(v.v[0]), b = code_narrow<24>(v.v[1]), c = code_narrow<24>(v.v[2]), d = code_narrow<24>(v.v[3]);
        p.w[0] = a | (b << 24); p.w[1] = (b >> 8) | (c << 16); p.w[2] = (c >> 16) | (d << 8);
    }
    return p;
}
__device__ __forceinline__ scr4 unpack4(const pk4 &p)
{
    scr4 v;
    if constexpr (CODE_BITS == 32) {
#pragma unroll
        for (int k = 0; k < 4; k++) v.v[k] = __uint_as_float(p.w[k]);
    } else if constexpr (CODE_BITS == 16) {
        v.v[0] = code_widen<16>(p.w[0] & 0xffffu); v.v[1] = code_widen<16>(p.w[0] >> 16);
        v.v[2] = code_widen<16>(p.w[1] & 0xffffu); v.v[3] = code_widen<16>(p.w[1] >> 16);
    } else {
        v.v[0] = code_widen<24>(p.w[0] & 0xffffffu);
        v.v[1] = code_widen<24>(__builtin_amdgcn_alignbit(p.w[1], p.w[0], 24) & 0xffffffu);
        v.v[2] = code_widen<24>(__builtin_amdgcn_alignbit(p.w[2], p.w[1], 16) & 0xffffffu);
        v.v[3] = code_widen<24>(p.w[2] >> 8);
    }
    return v;
}
typedef unsigned pk_vec __attribute__((ext_vector_type(CODE_WORDS)));
__device__ __forceinline__ void scr_store(unsigned *base, size_t cell, const scr4 &v)
{
    const pk4 p = pack4(v);
    pk_vec x;
#pragma unroll
    for (int k = 0; k < CODE_WORDS; k++) x[k] = p.w[k];
#ifdef RRLW_KO_STORES       // knock-out (timing only, wrong results): the codes are formed - the empty asm keeps every one alive - and not stored
    (void)base; (void)cell;
#pragma unroll
    for (int k = 0; k < CODE_WORDS; k++) asm volatile("" :: "v"(x[k]));
    return;
#endif
#ifdef RRLW_KO_HALFSTORE    // knock-out (timing only, wrong results): half of every record is stored - as many store instructions, half the bytes
    {
        typedef unsigned v2h __attribute__((ext_vector_type(2)));
        v2h lo = {p.w[0] ^ p.w[2], p.w[1] ^ p.w[3 % CODE_WORDS]};
        __builtin_nontemporal_store(lo, reinterpret_cast<v2h *>(base + cell * 2));       // (contiguous: whole lines are written)
        return;
    }
#endif
#ifdef RRLW_NO_NT
    __builtin_memcpy(base + cell * CODE_WORDS, &x, CODE_BYTES);
#else
    if constexpr (CODE_WORDS == 3) {        // (no 12-byte vector store through a pointer: 8 + 4)
        typedef unsigned v2 __attribute__((ext_vector_type(2)));
        v2 lo = {p.w[0], p.w[1]};
        __builtin_nontemporal_store(lo, reinterpret_cast<v2 *>(base + cell * 3));
        __builtin_nontemporal_store(p.w[2], base + cell * 3 + 2);
    } else {
        __builtin_nontemporal_store(x, reinterpret_cast<pk_vec *>(base + cell * CODE_WORDS));
    }
#endif
}
// transmittance-table index of an optical depth: rtrn :445 (tblint = 10000, Pade constant bpade)
__device__ __forceinline__ int lut_index(double od, double bpade) { return (int)(10000.0 * fdiv(od, bpade + od) + 0.5); }

// code of a series-branch cell: its optical depth as a float, negative values (and NaN) to zero - the clamp modifier of the conversion
// ([0, 1]; a thin cell is <= 0.06)
__device__ __forceinline__ scr_t thin_code(double od) { return __builtin_amdgcn_fmed3f((scr_t)od, 0.0f, 1.0f); }

// table indices of the four cells of a quad, formed together when any of them is thick (see gas_codes); a thin cell's index is not used
__device__ __forceinline__ void quad_indices(double o0, double o1, double o2, double o3, bool t0, bool t1, bool t2, bool t3, double bpade, int *ig)
{
    ig[0] = ig[1] = ig[2] = ig[3] = 0;
    if (t0 || t1 || t2 || t3) {
        ig[0] = lut_index(o0, bpade); ig[1] = lut_index(o1, bpade); ig[2] = lut_index(o2, bpade); ig[3] = lut_index(o3, bpade);
    }
}
__device__ __forceinline__ void quad_indices(double o0, double o1, double o2, double o3, double bpade, int *ig)
{
    quad_indices(o0, o1, o2, o3, !(o0 <= 0.06), !(o1 <= 0.06), !(o2 <= 0.06), !(o3 <= 0.06), bpade, ig);
}

// What k_layer hands to the sweeps for one cell is the DECISION the reference takes on its optical depth (rtrn :372-451), in 4 bytes:
//   code >= 0 : the optical depth itself (<= 0.06, or odtot < 0.06) - the series branch: atrans = od - od^2/2, tfac = od/6
//   code <  0 : -(index into the 1e-4-quantised transmittance / tfn tables), formed here in float64 exactly as the reference forms it
// The sweep turns the code back into (transmittance, tfn factor) and forms the Planck source terms in float64 itself.  Compared with
// storing {atrans, bbd, bbu} as three floats this is a third of the bytes, and the table index never depends on a rounded value.
__device__ __forceinline__ scr_t cell_code(double od, bool series, double bpade)
{
    return series ? (scr_t)od : -(scr_t)lut_index(od, bpade);
}

// all cells (g-points) of band B of one (layer, column).
// CLOUD: 0 clear-sky set, 1 one cloud optical depth per band (rtrn / rtrnmr), 2 one per g-point from W.odg (rtrnmc, sub-column
// arrays), 3 the band's value where the sub-column mask has a bit (rtrnmc, generator mask; gbits = the band's ng mask bits)
template <int B, int CLOUD, bool LOWER, int N, bool LDS>
__device__ __forceinline__ void band_cells(const DevTables &T, const Workspace &W, __amdgpu_buffer_rsrc_t kt, const double2 *lds, const Rows<N> &rw, int lay, int col,
                                           bool incol, double secdiff, double taua, int cloudy, double odcld, unsigned gbits)
{
    constexpr Region R = LOWER ? BT<B>::lo : BT<B>::up;
    constexpr int ng = BT<B>::ng;
    constexpr int NQ = band_nquad(B), NP = 4 * NQ, QS = band_qstart(B);
    const int nlay = W.nlay;
    const size_t ncb = W.ncolb;
    double od[NP];
    rows_eval_band<B, LOWER, N, LDS>(kt, lds, rw, taua, od);
    const double *S = T.stat;
    const double *__restrict__ tau_tbl = S + T.sl.tau_tbl;
    const double bpade = T.bpade;
    // (kept in the basic block of the table loads: an instruction-sinking pass would otherwise move the whole FMA chains
    // below the next branch and keep every loaded row alive until then)
#pragma unroll
    for (int j = 0; j < NP; j++) {        // optical depth along the diffusivity angle: rtrn :368-369
        // The reference's `if (odepth < 0) odepth = 0` is NOT applied here: a negative value is a thin cell, and its code is formed with
        // the conversion's clamp modifier (thin_code: one instruction, where a float64 compare and two selects per cell stood until round
        // 5; fmax(o, 0) - one v_max_f64 - makes the scheduler overlap more of the band and spills 29 registers: this kernel sits at its
        // 168-register budget).  The cloudy branch, which adds the cloud optical depth to this value, clamps it first (below).
        double o = secdiff * od[j];          // (od: gas + aerosol, rows_eval_band)
        if (j >= ng) o = 0.0;
        od[j] = o;
    }
#pragma unroll
    for (int j = 0; j < NP; j++) asm volatile("" : "+v"(od[j]));   // pin: the values exist here (LLVM's Sink pass may not move their FMA chains past this point)
    STAMP(3);
    // (the stores stay inside `if (incol)`: made unconditional - lanes past the last column write what the last column writes - the
    // scheduler moves code across them and spills 26 registers: 24.4 -> 31.5 ms)
    if constexpr (R.key == K_BINARY) { if (incol) W.fw[((size_t)fw_slot(B) * nlay + (lay - 1)) * ncb + col] = rw.fw; }
    const size_t so0 = ((size_t)QS * nlay + (lay - 1)) * ncb + col;     // scratch cell of the band's first quad
    const size_t qstride = (size_t)nlay * ncb;
    // gas: series for od <= 0.06, else the table (rtrn :372-451; in a cloudy layer the branch taken for the gas terms depends on od alone)
    auto gas_codes = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            scr4 c;
            // the table index costs a float64 division (~14 instructions per cell, a third of k_layer's arithmetic): formed for the four
            // cells of a QUAD together when any of them is thick (the g-points of a band are ordered by absorption: thick cells are
            // neighbours) and skipped for a wave whose quad is all thin (upper atmosphere, band wings).  One exec-mask bracket per quad
            // instead of one per cell - a quarter of the scalar instructions - and four independent division chains in flight instead
            // of one after the other (round 5: profiles/round5_instruction_diet.md).
            const bool t0 = !(od[4 * q] <= 0.06), t1 = !(od[4 * q + 1] <= 0.06), t2 = !(od[4 * q + 2] <= 0.06), t3 = !(od[4 * q + 3] <= 0.06);
#pragma unroll
            for (int k = 0; k < 4; k++) c.v[k] = thin_code(od[4 * q + k]);
            if (t0 || t1 || t2 || t3) {
                const scr_t i0 = -(scr_t)lut_index(od[4 * q], bpade), i1 = -(scr_t)lut_index(od[4 * q + 1], bpade);
                const scr_t i2 = -(scr_t)lut_index(od[4 * q + 2], bpade), i3 = -(scr_t)lut_index(od[4 * q + 3], bpade);
                c.v[0] = t0 ? i0 : c.v[0]; c.v[1] = t1 ? i1 : c.v[1]; c.v[2] = t2 ? i2 : c.v[2]; c.v[3] = t3 ? i3 : c.v[3];
            }
            if (incol) scr_store(W.scr[S_CODE], so0 + q * qstride, c);
            __builtin_amdgcn_sched_barrier(0);      // (one quad's chains at a time: all of a band's at once cost registers this kernel does not have)
        }
    };
    if constexpr (CLOUD == 0) {
        gas_codes();
    } else {
        // A wave with a cloudy lane (all 64 columns sit in the same layer) forms, per group of quads, the table index of the thick cells ONCE
        // and uses it for the gas code and for the cloudy branch p3 (round 3: the two passes of round 2 divided twice; a cloudy layer cost
        // 3.2 x a clear one).  The three sub-branches of rtrn :372-435:
        //   p1: odtot < 0.06              total from the series
        //   p2: else if odepth <= 0.06    total = odepth + odcld from the table
        //   p3: else                      odepth := tau_tbl(itgas), total from the table
        // (McICA, CLOUD 2 / 3: every g-point has its own cloud optical depth - eight more live doubles per quad in flight; with two quads
        // the kernel spilled 13-18 registers)
        // (the per-g-point flavours keep the two passes: with the index held across the group they spilled 7-9 registers)
        constexpr bool REUSE = CLOUD == 1;
        const bool wave_cloudy = __builtin_amdgcn_ballot_w64(cloudy != 0) != 0ull;
        if (!REUSE || !wave_cloudy) gas_codes();
        if (wave_cloudy) {
#pragma unroll
            for (int j = 0; j < ng; j++) { if (!(od[j] >= 0.0)) od[j] = 0.0; }       // rtrn :369 (see above)
            constexpr int QCW = CLOUD >= 2 ? 1 : RRLW_CLOUD_QUADS;
            constexpr int QC = QCW < NQ ? QCW : NQ;
#pragma unroll
            for (int q0 = 0; q0 < NQ; q0 += QC) {
                constexpr int GC = 4 * QC;
                int ig[REUSE ? GC : 1];     // table index of the gas optical depth, 0 for a series cell
                if constexpr (REUSE) {
#pragma unroll
                    for (int q = 0; q < QC; q++) {
                        const int j = 4 * (q0 + q);         // (padded cells hold od = 0: thin)
                        if (q0 + q < NQ) quad_indices(od[j < NP ? j : 0], od[j + 1 < NP ? j + 1 : 0], od[j + 2 < NP ? j + 2 : 0], od[j + 3 < NP ? j + 3 : 0], bpade, ig + 4 * q);
                        else ig[4 * q] = ig[4 * q + 1] = ig[4 * q + 2] = ig[4 * q + 3] = 0;
                    }
#pragma unroll
                    for (int q = 0; q < QC; q++) {
                        if (q0 + q < NQ) {
                            scr4 c;
#pragma unroll
                            for (int kk = 0; kk < 4; kk++) {
                                const int j = 4 * (q0 + q) + kk;
                                c.v[kk] = od[j < NP ? j : 0] <= 0.06 ? (scr_t)od[j < NP ? j : 0] : -(scr_t)ig[4 * q + kk];
                            }
                            if (incol) scr_store(W.scr[S_CODE], so0 + (q0 + q) * qstride, c);
                        }
                    }
                }
                if (cloudy) {
                    double odc[GC], tg[GC];
#pragma unroll
                    for (int k = 0; k < GC; k++) odc[k] = odcld;
                    if constexpr (CLOUD == 3) {
#pragma unroll
                        for (int k = 0; k < GC; k++) odc[k] = ((gbits >> (4 * q0 + k)) & 1u) ? odcld : 0.0;
                    }
                    if constexpr (CLOUD == 2) {
#pragma unroll
                        for (int q = 0; q < QC; q++) {
                            if (q0 + q < NQ) {
                                const double2 *po = reinterpret_cast<const double2 *>(W.odg + (so0 + (q0 + q) * qstride) * 4);
                                const double2 a = po[0], b = po[1];
                                odc[4 * q] = a.x; odc[4 * q + 1] = a.y; odc[4 * q + 2] = b.x; odc[4 * q + 3] = b.y;
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < GC; k++) {
                        const int j = 4 * q0 + k;
                        if (j < NP) {
                            const bool p3 = !(od[j] + odc[k] < 0.06) && !(od[j] <= 0.06);
                            int igk = 0;
                            if constexpr (REUSE) igk = p3 ? ig[k] : 0;
                            else { if (j < ng && p3) igk = lut_index(od[j], bpade); }
                            tg[k] = tau_tbl[igk];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < QC; q++) {
                        if (q0 + q < NQ) {
                            scr4 c;
                            double odtot[4];
                            bool tk[4];             // the total optical depth takes the table (a padded cell never does)
                            int it[4];
#pragma unroll
                            for (int kk = 0; kk < 4; kk++) {
                                const int k = 4 * q + kk, j = 4 * (q0 + q) + kk;
                                const bool p1 = od[j] + odc[k] < 0.06;
                                const bool p3 = !p1 && !(od[j] <= 0.06);
                                odtot[kk] = (p3 ? tg[k] : od[j]) + odc[k];
                                tk[kk] = j < ng && !p1;
                            }
                            quad_indices(odtot[0], odtot[1], odtot[2], odtot[3], tk[0], tk[1], tk[2], tk[3], bpade, it);
#pragma unroll
                            for (int kk = 0; kk < 4; kk++) {
                                const int j = 4 * (q0 + q) + kk;
                                c.v[kk] = j < ng ? (tk[kk] ? -(scr_t)it[kk] : (scr_t)odtot[kk]) : (scr_t)0;
                            }
                            if (incol) scr_store(W.scr[S_CODET], so0 + (q0 + q) * qstride, c);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        }
    }
}

#ifndef RRLW_LAYER_WAVES
#define RRLW_LAYER_WAVES 3        // waves per SIMD k_layer is compiled for (168 VGPRs; three workgroups of 256 threads and 53 KB of LDS per CU)
#endif

// workgroup-level state of the LDS staging: the staged region (lower / upper atmosphere), first pressure plane and first minor-gas
// temperature slice, this thread's place in the copy loops, and whether this thread's cell lies inside the staged window
struct LayerWg { double2 *lds; int jp0, im0, tid, nth, pc; bool lower, ok; };      // pc: the thread's column of the batch (caller arrays; `col` is its position)

// all cells of band B for one (layer, column); BASE_LO / BASE_UP: where the band's tables start in the staging buffer of its pass
template <int B, int CLOUD, class WN, int BASE_LO, int BASE_UP>
__device__ __forceinline__ void layer_band(const DevTables &T, const Workspace &W, const LayerArgs &a, const LayerCoef &C,
                                           __amdgpu_buffer_rsrc_t kt, const LayerWg &wg, bool lower, int lay, int col, bool incol, int cloudy,
                                           const unsigned (&mw)[MASK_WORDS])
{
    STAMP(4);                                                                   // (cells of the previous band)
    unsigned delta[NROLE];
#pragma unroll
    for (int r = 0; r < NROLE; r++) delta[r] = 0u;
    const bool use_lds = __builtin_amdgcn_ballot_w64(!wg.ok) == 0ull;       // wave-uniform: every cell of the wave lies in the staged window
    const size_t ncb = W.ncolb, gcx = (size_t)a.col0 + wg.pc;
    double odcld = 0.0;
    if constexpr (CLOUD == 1 || CLOUD == 3) { if (cloudy) odcld = W.odcld[((size_t)(B - 1) * W.nlay + (lay - 1)) * ncb + col]; }   // (written for cloudy layers only)
    const double secdiff = W.percol[(size_t)(PC_SECDIFF + B - 1) * ncb + col];
    const double taua = col_load(a.tauaer + (size_t)a.col0 + (size_t)a.nct * ((lay - 1) + (size_t)W.nlay * (B - 1)), (unsigned)wg.pc * 8u);
    unsigned gbits = 0u;
    if constexpr (CLOUD == 3) {
        // the band's bits of the sub-column mask: the one or two words that hold them, read here (all five held from the top of the kernel
        // ended up in scratch memory)
        (void)mw;
        constexpr int g0 = band_g0(B), ng = BT<B>::ng, w0 = g0 >> 5, w1 = (g0 + ng - 1) >> 5;
        if (cloudy) {
            const unsigned *mrow = W.mask + (size_t)(lay - 1) * W.mask_stride + W.mask_col0 + gcx;
            const unsigned long long lo = mrow[(size_t)w0 * W.nlay * W.mask_stride];
            const unsigned long long hi = w1 != w0 ? mrow[(size_t)w1 * W.nlay * W.mask_stride] : 0u;
            gbits = (unsigned)(((lo | (hi << 32)) >> (g0 & 31)) & ((1ull << ng) - 1ull));
        }
    }
    if (lower) {
        constexpr int N = region_nrows(BT<B>::lo, true);
        Rows<N> rw;
        rows_prep<B, true, N>(T, C, rw);
        if (use_lds) {
            band_delta<B, true, WN, BASE_LO>(T, wg.jp0, wg.im0, delta);
            rows_to_lds<B, true, N>(rw, delta);
            STAMP(2);
            band_cells<B, CLOUD, true, N, true>(T, W, kt, wg.lds, rw, lay, col, incol, secdiff, taua, cloudy, odcld, gbits);
        } else {
            band_cells<B, CLOUD, true, N, false>(T, W, kt, wg.lds, rw, lay, col, incol, secdiff, taua, cloudy, odcld, gbits);
        }
    } else {
        constexpr int N = region_nrows(BT<B>::up, false);
        Rows<N> rw;
        rows_prep<B, false, N>(T, C, rw);
        if (use_lds) {
            band_delta<B, false, WN, BASE_UP>(T, wg.jp0, wg.im0, delta);
            rows_to_lds<B, false, N>(rw, delta);
            STAMP(2);
            band_cells<B, CLOUD, false, N, true>(T, W, kt, wg.lds, rw, lay, col, incol, secdiff, taua, cloudy, odcld, gbits);
        } else {
            band_cells<B, CLOUD, false, N, false>(T, W, kt, wg.lds, rw, lay, col, incol, secdiff, taua, cloudy, odcld, gbits);
        }
    }
}

// The staging passes: every band once, in evaluation order (bands that are the only users of some setcoef quantity - pavel, broadening
// gases, O2, halocarbons, CO - first, the bands with the longest row lists last: their register peak then meets fewer live quantities).
// A pass must fit the staging buffer in both regions of the atmosphere (static_assert in stage_pass).
#ifdef RRLW_G256
using LayerPasses = std::tuple<BandList<1, 2, 11, 15, 6>, BandList<8, 10, 14, 16>, BandList<12>, BandList<13>, BandList<4, 9>, BandList<7>, BandList<3>, BandList<5>>;
#elif defined(RRLW_LAYER_PASS_PER_BAND)       // measurement: one staging round per band, as before round 3
using LayerPasses = std::tuple<BandList<1>, BandList<2>, BandList<11>, BandList<15>, BandList<6>, BandList<8>, BandList<10>, BandList<14>, BandList<16>,
                               BandList<12>, BandList<13>, BandList<4>, BandList<9>, BandList<7>, BandList<3>, BandList<5>>;
#elif defined(RRLW_LAYER_PASSES_3)              // measurement: 384-thread workgroups with a 9 000-double buffer (two per CU)
using LayerPasses = std::tuple<BandList<1, 2, 11, 15, 6, 8, 10, 14, 16, 12, 13, 4>, BandList<9, 7, 3>, BandList<5>>;
#elif defined(RRLW_LAYER_PASSES_1)              // measurement: one 768-thread workgroup per CU, every table of the region staged at once
using LayerPasses = std::tuple<BandList<1, 2, 11, 15, 6, 8, 10, 14, 16, 12, 13, 4, 9, 7, 3, 5>>;
#else
using LayerPasses = std::tuple<BandList<1, 2, 11, 15, 6, 8, 10, 14, 16, 12, 13>, BandList<4, 9>, BandList<7, 3>, BandList<5>>;
#endif
// ... and with the wide window (five planes, ten minor-gas slices: the tables of the binary-key bands take 5/3 of the space)
#ifdef RRLW_G256
using LayerPassesWide = LayerPasses;
#else
using LayerPassesWide = std::tuple<BandList<1, 2, 11, 15, 6, 8, 10, 14, 16, 13>, BandList<12, 4>, BandList<9>, BandList<7>, BandList<3>, BandList<5>>;
#endif

template <class PL, class WN, int CLOUD, int... I>
__device__ __forceinline__ void pass_run(std::integer_sequence<int, I...>, const DevTables &T, const Workspace &W, const LayerArgs &a,
                                         const LayerCoef &C, __amdgpu_buffer_rsrc_t kt, const LayerWg &wg, bool lower, int lay, int col, bool incol,
                                         int cloudy, const unsigned (&mw)[MASK_WORDS], unsigned bmask)
{
    // (a workgroup that holds none of the pass's bands - the bands of a small batch spread over several - leaves the pass alone: uniform)
    constexpr unsigned passmask = ((1u << (PL::b[I] - 1)) | ...);
    if ((bmask & passmask) == 0u) return;
    // the pass's tables -> LDS; the barriers are reached by every thread of the workgroup
    STAMP(4);
    lds_barrier();                      // the previous pass's readers are done with the staging buffer
    STAMP(0);
    if (wg.lower) stage_pass<PL, true, WN>(T, kt, wg.lds, wg.jp0, wg.im0, wg.tid);
    else stage_pass<PL, false, WN>(T, kt, wg.lds, wg.jp0, wg.im0, wg.tid);
    lds_barrier();
    STAMP(1);
    (((bmask >> (PL::b[I] - 1)) & 1u
          ? layer_band<PL::b[I], CLOUD, WN, pass_base<PL, true, WN, I>(), pass_base<PL, false, WN, I>()>(T, W, a, C, kt, wg, lower, lay, col, incol, cloudy, mw)
          : (void)0), ...);
}
template <int CLOUD, class WN, class... PLs>
__device__ __forceinline__ void passes_run(std::tuple<PLs...> *, const DevTables &T, const Workspace &W, const LayerArgs &a,
                                           const LayerCoef &C, __amdgpu_buffer_rsrc_t kt, const LayerWg &wg, bool lower, int lay, int col, bool incol,
                                           int cloudy, const unsigned (&mw)[MASK_WORDS], unsigned bmask)
{
    (pass_run<PLs, WN, CLOUD>(std::make_integer_sequence<int, PLs::n>{}, T, W, a, C, kt, wg, lower, lay, col, incol, cloudy, mw, bmask), ...);
}

// One workgroup's work: the cells of window `bx` (256 consecutive positions) in layer `by` + 1.  WIDE = 0: the narrow staging window; a
// workgroup whose cells do not fit it leaves its (window, layer) in W.wide and returns.  WIDE = 1: the wide window (k_layer<.., 1> walks
// that list).
template <bool GCM, int CLOUD, int WIDE>
__device__ __forceinline__ void layer_cells(const DevTables &T, const Workspace &W, const GcmIn &g, const ColIn &c, const LayerArgs &a, double2 *s_tab, int *s_wg,
                                            int bx, int by, unsigned bmask, bool lists)
{
    const int colr = bx * LAYER_BLOCK + threadIdx.x;
    const bool incol = colr < a.ncol;
    const int col = incol ? colr : a.ncol - 1;      // threads past the end shadow the last column (they take part in the staging and the barriers)
    const int lay = by + 1;
    const int pc = pcol(W, col);                    // the column this position holds (k_colsort)
    const size_t gc = (size_t)a.col0 + pc;
    const int nct = a.nct;
    const size_t gi = gc + (size_t)nct * (lay - 1);
    const double *S = T.stat;
    const double *preflog = S + T.sl.preflog, *tref = S + T.sl.tref;
    const double amd = 28.9660, amw = 18.0160, avogad = 6.02214199e+23, grav = 9.8066;
    const double stpfac = 296. / 1013.;

    // The narrow launch decides FIRST whether the workgroup's cells fit its window - from the layer's pressures and temperatures alone -
    // and a workgroup that leaves for the wide launch has then read two of its fifteen input rows and computed nothing else (on a
    // terrain-following grid nine workgroups in ten leave: profiles/round5_orography.md).  The others go on as before; what they have found
    // of the window stays in s_wg.
    bool early = false;
    if constexpr (WIDE == 0 && GCM && HAVE_WIDE) {
        if (W.wide) {           // (uniform)
            early = true;
            const size_t ro = (size_t)a.col0 + (size_t)nct * (lay - 1);
            const double pe = col_load(g.play + ro, (unsigned)pc * 8u), te = col_load(g.tlay + ro, (unsigned)pc * 8u);
            const int jpe = clampi((int)(36. - 5 * (log(pe) + 0.04)), 1, 58);
            const int ime = min(18, max(1, (int)fdiv(te - 180.8, 7.2)));
            const bool lowe = lay <= W.laytrop[col];
            if (threadIdx.x == 0) { s_wg[0] = lowe ? 1 : 0; s_wg[1] = 99; s_wg[2] = 99; s_wg[3] = 0; s_wg[4] = 0; }
            __syncthreads();
            const bool in = lowe == (s_wg[0] != 0);
            int jlo = in ? jpe : 99, ilo = in ? ime : 99, jhi = in ? jpe : 0, ihi = in ? ime : 0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                jlo = min(jlo, __shfl_xor(jlo, o, 64)); ilo = min(ilo, __shfl_xor(ilo, o, 64));
                jhi = max(jhi, __shfl_xor(jhi, o, 64)); ihi = max(ihi, __shfl_xor(ihi, o, 64));
            }
            if ((threadIdx.x & 63) == 0) { atomicMin(&s_wg[1], jlo); atomicMin(&s_wg[2], ilo); atomicMax(&s_wg[3], jhi); atomicMax(&s_wg[4], ihi); }
            __syncthreads();
            const int jp0 = s_wg[1], im0 = min(s_wg[2], 20 - WinNarrow::MW);
            if (s_wg[3] - jp0 > WinNarrow::NPL - 2 || s_wg[4] - im0 > WinNarrow::MW - 2) {
                if (threadIdx.x == 0 && lists) W.wide[2 + atomicAdd(&W.wide[0], 1)] = by * ((a.ncol + LAYER_BLOCK - 1) / LAYER_BLOCK) + bx;     // (one of the pair's workgroups lists it)
                return;
            }
        }
    }

    double pavel, tavel, coldry, wbrodl, w1, w2, w3, w4, w5, w6, w7;
    LayerCoef C;
    if (GCM) {
        const size_t ro = (size_t)a.col0 + (size_t)nct * (lay - 1);      // the layer's row of a (ncol, nlay) array (uniform)
        const unsigned o8 = (unsigned)pc * 8u;
        pavel = col_load(g.play + ro, o8);
        tavel = col_load(g.tlay + ro, o8);
        const double pz_lo = col_load(g.plev + ro, o8), pz_hi = col_load(g.plev + ro + nct, o8);
        w1 = col_load(g.h2ovmr + ro, o8); w2 = col_load(g.co2vmr + ro, o8); w3 = col_load(g.o3vmr + ro, o8); w4 = col_load(g.n2ovmr + ro, o8);
        w5 = 0.0; w6 = col_load(g.ch4vmr + ro, o8); w7 = col_load(g.o2vmr + ro, o8);
        const double amm = (1. - w1) * amd + w1 * amw;
        coldry = fdiv((pz_lo - pz_hi) * 1.e3 * avogad, 1.e2 * grav * amm * (1. + w1));
        double summol = 0.0;
        summol = summol + w2; summol = summol + w3; summol = summol + w4;
        summol = summol + w5; summol = summol + w6; summol = summol + w7;
        wbrodl = coldry * (1. - summol);
        w1 = coldry * w1; w2 = coldry * w2; w3 = coldry * w3; w4 = coldry * w4;
        w5 = coldry * w5; w6 = coldry * w6; w7 = coldry * w7;
        C.f[F_WX1] = coldry * col_load(g.ccl4vmr + ro, o8) * 1.e-20;
        C.f[F_WX2] = coldry * col_load(g.cfc11vmr + ro, o8) * 1.e-20;
        C.f[F_WX3] = coldry * col_load(g.cfc12vmr + ro, o8) * 1.e-20;
        C.f[F_WX4] = coldry * col_load(g.cfc22vmr + ro, o8) * 1.e-20;
    } else {
        pavel = c.pavel[gi];
        tavel = c.tavel[gi];
        coldry = c.coldry[gi];
        wbrodl = c.wbrodl[gi];
        const size_t wi = gc + (size_t)nct * 7 * (lay - 1);       // wkl (ncol,7,nlayers)
        w1 = c.wkl[wi]; w2 = c.wkl[wi + (size_t)nct]; w3 = c.wkl[wi + (size_t)nct * 2];
        w4 = c.wkl[wi + (size_t)nct * 3]; w5 = c.wkl[wi + (size_t)nct * 4]; w6 = c.wkl[wi + (size_t)nct * 5];
        w7 = c.wkl[wi + (size_t)nct * 6];
        const size_t xi = gc + (size_t)nct * 4 * (lay - 1);       // wx (ncol,4,nlayers)
        C.f[F_WX1] = c.wx[xi]; C.f[F_WX2] = c.wx[xi + (size_t)nct];
        C.f[F_WX3] = c.wx[xi + (size_t)nct * 2]; C.f[F_WX4] = c.wx[xi + (size_t)nct * 3];
    }

    // (the Planck functions of setcoef :173-269 are formed by the sweeps, which need them per band and level)
    // pressure / temperature interpolation: setcoef :276-306
    const double plog = log(pavel);
    const int jp = clampi((int)(36. - 5 * (plog + 0.04)), 1, 58);
    const double fp = 5. * (preflog[jp - 1] - plog);
    const double dt0 = fdiv(tavel - tref[jp - 1], 15.), dt1 = fdiv(tavel - tref[jp], 15.);
    const int jt = clampi((int)(3. + dt0), 1, 4);
    const double ft = dt0 - (double)(jt - 3);
    const int jt1 = clampi((int)(3. + dt1), 1, 4);
    const double ft1 = dt1 - (double)(jt1 - 3);
    const double water = fdiv(w1, coldry);
    const double scalefac = fdiv(pavel * stpfac, tavel);
    int indself = 1, indfor, indminor;
    double forfac, forfrac, selffac, selffrac = 0.0, factor;
    if (!(plog <= 4.56)) {            // :312-334
        forfac = fdiv(scalefac, 1. + water);
        factor = fdiv(332.0 - tavel, 36.0);
        indfor = min(2, max(1, (int)factor));
        forfrac = factor - (double)indfor;
        selffac = water * forfac;
        factor = fdiv(tavel - 188.0, 7.2);
        indself = min(9, max(1, (int)factor - 7));
        selffrac = factor - (double)(indself + 7);
    } else {                          // :369-377
        forfac = fdiv(scalefac, 1. + water);
        factor = fdiv(tavel - 188.0, 36.0);
        indfor = 3;
        forfrac = factor - 1.0;
        selffac = water * forfac;
    }
    factor = fdiv(tavel - 180.8, 7.2);
    indminor = min(18, max(1, (int)factor));
    C.f[F_MINORFRAC] = factor - (double)indminor;
    C.f[F_SCALEMINOR] = fdiv(pavel, tavel);
    C.f[F_SCALEMINORN2] = C.f[F_SCALEMINOR] * fdiv(wbrodl, coldry + w1);
    double colh2o = 1.e-20 * w1, colco2 = 1.e-20 * w2, colo3 = 1.e-20 * w3, coln2o = 1.e-20 * w4;
    double colco = 1.e-20 * w5, colch4 = 1.e-20 * w6, colo2 = 1.e-20 * w7;
    if (colco2 == 0.) colco2 = 1.e-32 * coldry;
    if (colo3 == 0.) colo3 = 1.e-32 * coldry;
    if (coln2o == 0.) coln2o = 1.e-32 * coldry;
    if (colco == 0.) colco = 1.e-32 * coldry;
    if (colch4 == 0.) colch4 = 1.e-32 * coldry;
    const double compfp = 1. - fp;     // :421-429
    C.f[F_FAC10] = compfp * ft;
    C.f[F_FAC00] = compfp * (1. - ft);
    C.f[F_FAC11] = fp * ft1;
    C.f[F_FAC01] = fp * (1. - ft1);
    C.f[F_SELFFAC] = colh2o * selffac;
    C.f[F_FORFAC] = colh2o * forfac;
    C.f[F_SELFFRAC] = selffrac;
    C.f[F_FORFRAC] = forfrac;
    C.f[F_COLH2O] = colh2o; C.f[F_COLCO2] = colco2; C.f[F_COLO3] = colo3; C.f[F_COLN2O] = coln2o;
    C.f[F_COLCO] = colco; C.f[F_COLCH4] = colch4; C.f[F_COLO2] = colo2; C.f[F_COLBRD] = 1.e-20 * wbrodl;
    C.f[F_COLDRY] = coldry;
    C.f[F_PAVEL] = pavel;
    C.jp = jp; C.jt = jt; C.jt1 = jt1; C.indself = indself; C.indfor = indfor; C.indminor = indminor;

    // "lower atmosphere" for taumol is lay <= laytrop (the count of layers with ln p > 4.56), :312-313
    const bool lower = lay <= W.laytrop[col];
    int cloudy = 0;
    // (threads past the last column never take the cloudy branches: nothing of theirs is stored, and with them no lane of a divergent
    // region skips the `if (incol)` at its end - the construct of profiles/round5_exec_hazard.md)
    if (CLOUD) cloudy = incol ? (W.cflag[(size_t)lay * W.ncolb + col] & 1) : 0;
    // wave-uniform descriptor of the packed k tables (built from kernel arguments only)
    const __amdgpu_buffer_rsrc_t kt = __builtin_amdgcn_make_buffer_rsrc((void *)T.ktab, 0, a.ktab_bytes, 0x00020000);
    const unsigned mw[MASK_WORDS] = {};
#ifdef RRLW_LAYER_STAMPS
    if ((threadIdx.x & 63) == 0) { for (int i = 0; i < NSTAMP; i++) s_stamp[(threadIdx.x >> 6) * (NSTAMP + 1) + i] = 0ull; }
    STAMP(-1);
#endif
    // staging window of the workgroup: the region of its first thread, the smallest jp and the smallest indminor among the cells of that region
    for (int i = threadIdx.x; i < NRATCHI; i += LAYER_BLOCK) s_ratchi[i] = S[T.sl.rat + i];
    if (!early) {
        if (threadIdx.x == 0) { s_wg[0] = lower ? 1 : 0; s_wg[1] = 99; s_wg[2] = 99; s_wg[3] = 0; s_wg[4] = 0; }
        __syncthreads();
    }
    LayerWg wg;
    wg.lds = s_tab; wg.tid = threadIdx.x; wg.nth = blockDim.x; wg.pc = pc;
    wg.lower = s_wg[0] != 0;
    // (one wave-level reduction per quantity, then one LDS atomic per wave: 256 atomics on one address serialise)
    if (!early) {
        const bool in = lower == wg.lower;
        int jlo = in ? jp : 99, ilo = in ? indminor : 99, jhi = in ? jp : 0, ihi = in ? indminor : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            jlo = min(jlo, __shfl_xor(jlo, o, 64)); ilo = min(ilo, __shfl_xor(ilo, o, 64));
            jhi = max(jhi, __shfl_xor(jhi, o, 64)); ihi = max(ihi, __shfl_xor(ihi, o, 64));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_wg[1], jlo); atomicMin(&s_wg[2], ilo); atomicMax(&s_wg[3], jhi); atomicMax(&s_wg[4], ihi); }
    }
    __syncthreads();
    wg.jp0 = s_wg[1];
    using WN = std::conditional_t<WIDE != 0, WinWide, WinNarrow>;
    wg.im0 = min(s_wg[2], 20 - WN::MW);                     // (the window ends with the table's last slice at the latest)
    wg.ok = lower == wg.lower && (unsigned)(jp - wg.jp0) <= (unsigned)(WN::NPL - 2) && (unsigned)(indminor - wg.im0) <= (unsigned)(WN::MW - 2);
    STAMP(5);                       // set-up of the workgroup's staging window
    passes_run<CLOUD, WN>(static_cast<std::conditional_t<WIDE != 0, LayerPassesWide, LayerPasses> *>(nullptr), T, W, a, C, kt, wg, lower, lay, col, incol, cloudy, mw, bmask);
#ifdef RRLW_LAYER_STAMPS
    STAMP(4);
    if ((threadIdx.x & 63) == 0) {
        for (int i = 0; i < NSTAMP; i++) atomicAdd(&g_stamps[i], s_stamp[(threadIdx.x >> 6) * (NSTAMP + 1) + i]);
        atomicAdd(&g_stamps[NSTAMP], 1ull);
    }
#endif
}

// WIDE = 0: grid (windows, layers), one workgroup per (window, layer).  WIDE = 1: grid (windows x layers); workgroup i takes pair i of the
// list the narrow launch left in W.wide = {count, -, pair ..} and leaves at once when there is none (a (window, layer) grid of
// workgroups that read one word and leave costs ~25 us per 250 000 columns; a fixed grid of workgroups that LOOP over the list spills
// 151 registers: everything derived from the kernel's arguments stays live round the loop).  The count is cleared on the stream in front
// of the narrow launch (driver.hip: run_layer).
template <bool GCM, int CLOUD, int WIDE>
__global__ __launch_bounds__(LAYER_BLOCK, RRLW_LAYER_WAVES) void k_layer(DevTables T, Workspace W, GcmIn g, ColIn c, LayerArgs a)
{
    __shared__ double2 s_tab[STAGE_DOUBLES / 2];
    __shared__ int s_wg[5];
    // the bands this workgroup takes: those of the call's range, of its part where the bands of a small batch are spread over several workgroups
    const int part = WIDE == 0 ? blockIdx.z : blockIdx.y;
    unsigned bmask = ((a.iend >= 32 ? 0u : (1u << a.iend)) - 1u) & ~((1u << (a.istart - 1)) - 1u);
    if (a.nparts > 1) bmask &= a.partmask[part];
    if constexpr (WIDE == 0) {
        layer_cells<GCM, CLOUD, 0>(T, W, g, c, a, s_tab, s_wg, blockIdx.x, blockIdx.y, bmask, part == 0);
    } else {
        const int n = __builtin_amdgcn_readfirstlane(W.wide[0]), gx = (a.ncol + LAYER_BLOCK - 1) / LAYER_BLOCK;
        if ((int)blockIdx.x >= n) return;
        const int item = __builtin_amdgcn_readfirstlane(W.wide[2 + blockIdx.x]);
        layer_cells<GCM, CLOUD, 1>(T, W, g, c, a, s_tab, s_wg, item % gx, item / gx, bmask, false);
    }
}

// ------------------------------------------------------------------------------------------------
// k_cloudmc : McICA cloud terms per g-point: cldprmc (src/rrtmg_lw_cldprmc.f90:49-279) followed by the cloud
//             part of rtrnmc's set-up (src/rrtmg_lw_rtrnmc.f90:307-329).  One thread per (column, layer).
//   FROMMASK = false: the sub-column arrays of the reference's McICA argument list, (140,ncol,nlay)
//                     (src/rrtmg_lw_rad.f90:267-298)
//   FROMMASK = true : sub-columns given by the generator's bit mask (k_subcol_*) plus the grid-mean cloud
//                     properties - the same values the generator would have expanded into those arrays
//                     (src/mcica_subcol_gen_lw.f90:664-680).
// ------------------------------------------------------------------------------------------------

template <bool FROMMASK>
__global__ __launch_bounds__(256) void k_cloudmc(DevTables T, Workspace W, McIn m, GcmIn g, int ncol, int col0, int nct,
                                                 int inflag, int iceflag, int liqflag)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    const int lay = blockIdx.y + 1;
    const int nlay = W.nlay;
    const size_t ncb = W.ncolb;
    const size_t gc = (size_t)col0 + pcol(W, col);
    const size_t cl = gc + (size_t)nct * (lay - 1);
    const double *S = T.stat;
    const double *absice1 = S + T.sl.absice1, *absice2 = S + T.sl.absice2, *absice3 = S + T.sl.absice3, *absliq1 = S + T.sl.absliq1;
    const double cldmin = 1.e-20;
    const double radice = FROMMASK ? g.reice[cl] : m.reicmcl[cl];
    const double radliq = FROMMASK ? g.reliq[cl] : m.relqmcl[cl];
    double mciwp = 0.0, mclwp = 0.0;
    unsigned mw[MASK_WORDS] = {};
    if (FROMMASK) {
        mciwp = g.cicewp[cl];
        mclwp = g.cliqwp[cl];
#pragma unroll
        for (int w = 0; w < MASK_WORDS; w++) mw[w] = W.mask[((size_t)w * nlay + (lay - 1)) * W.mask_stride + W.mask_col0 + gc];
    }
    if constexpr (FROMMASK) {
        unsigned anybit = 0u;
#pragma unroll
        for (int w = 0; w < MASK_WORDS; w++) anybit |= (w == MASK_WORDS - 1 && (NGPT & 31)) ? (mw[w] & ((1u << (NGPT & 31)) - 1u)) : mw[w];
        if (anybit == 0u) {       // no cloudy sub-column in this layer (140 = 4 x 32 + 12 bits)
            W.cflag[(size_t)lay * ncb + col] = 0;      // (odcld / efcl of a layer are read only when its flag is set)
            if (lay == 1) W.cflag[(size_t)(nlay + 1) * ncb + col] = 0;
            return;
        }
    }
    // particle-size interpolation positions (band independent): cldprmc :208-262
    int ice_err = 0, liq_err = 0, index_i = 1, index_l = 1;
    double fint_i = 0.0, fint_l = 0.0;
    if (iceflag == 0) { if (radice < 10.0) ice_err = E_ICE_SMALL; }
    else if (iceflag == 1) { if (radice < 13.0 || radice > 130.) ice_err = E_ICE_BOUNDS; }
    else if (iceflag == 2) {
        if (radice < 5.0 || radice > 131.0) ice_err = E_ICE_BOUNDS;
        const double factor = (radice - 2.) / 3.;
        index_i = (int)factor; if (index_i == 43) index_i = 42;
        fint_i = factor - (double)index_i;
        index_i = clampi(index_i, 1, 42);
    } else if (iceflag == 3) {
        if (radice < 5.0 || radice > 140.0) ice_err = E_ICE_GEN_BOUNDS;
        const double factor = (radice - 2.) / 3.;
        index_i = (int)factor; if (index_i == 46) index_i = 45;
        fint_i = factor - (double)index_i;
        index_i = clampi(index_i, 1, 45);
    }
    if (liqflag == 1) {
        if (radliq < 2.5 || radliq > 60.) liq_err = E_LIQ_BOUNDS;
        index_l = (int)(radliq - 1.5);
        if (index_l == 0) index_l = 1;
        if (index_l == 58) index_l = 57;
        fint_l = radliq - 1.5 - (double)index_l;
        index_l = clampi(index_l, 1, 57);
    }
    int err = 0, any = 0, quad = 0;
#pragma unroll 1
    for (int B = 1; B <= NBND; B++) {
        const int ng = band_ng(B), g0 = band_g0(B);
        const double secdiff = W.percol[(size_t)(PC_SECDIFF + B - 1) * ncb + col];
        double ai = 0.0, al = 0.0;         // coefficients of a cell of this band that holds ice / liquid
        if (inflag == 2) {
            if (iceflag == 0) ai = T.absice0[0] + T.absice0[1] / radice;
            else if (iceflag == 1) {
                const int icx = B <= 2 ? B : (B <= 5 ? 3 : (B <= 8 ? 4 : 5));                 // ipat(1:16,1), cldprmc :214-216
                ai = absice1[2 * (icx - 1)] + absice1[2 * (icx - 1) + 1] / radice;
            } else if (iceflag == 2) { const double *t = absice2 + 43 * (B - 1); ai = t[index_i - 1] + fint_i * (t[index_i] - t[index_i - 1]); }
            else if (iceflag == 3) { const double *t = absice3 + 46 * (B - 1); ai = t[index_i - 1] + fint_i * (t[index_i] - t[index_i - 1]); }
            if (liqflag == 0) al = T.absliq0;
            else if (liqflag == 1) { const double *t = absliq1 + 58 * (B - 1); al = t[index_l - 1] + fint_l * (t[index_l] - t[index_l - 1]); }
        }
        const double tband = FROMMASK ? g.taucld[(B - 1) + (size_t)NBND * cl] : 0.0;
        if constexpr (FROMMASK) {
            // every cloudy cell of this band and layer carries the same values (src/mcica_subcol_gen_lw.f90:664-680): one
            // optical depth / emissivity per band; k_layer<..,3,..> and k_sweepz<.,4> combine them with the mask bits
            const unsigned long long lo = mw[g0 >> 5], hi = (g0 >> 5) < MASK_WORDS - 1 ? mw[(g0 >> 5) + 1] : 0u;
            const unsigned bits = (unsigned)(((lo | (hi << 32)) >> (g0 & 31)) & ((1ull << ng) - 1ull));
            double t = tband, od = 0.0, ef = 0.0;
            if (bits) {
                if (mciwp + mclwp >= cldmin || t >= cldmin) {                                 // cldprmc :181-183 with cldfmc = 1
                    if (inflag == 1) err = E_MC_INFLAG1;
                    else if (inflag == 2) {
                        double a_i = 0.0, a_l = 0.0;
                        if (mciwp != 0.0 && iceflag >= 0 && iceflag <= 3) { a_i = ai; if (ice_err) err = ice_err; }
                        if (mclwp != 0.0) {
                            if (liqflag == 0) a_l = al;
                            else if (liqflag == 1) { a_l = al; if (liq_err) err = liq_err; }
                        }
                        t = mciwp * a_i + mclwp * a_l;
                    }
                }
                od = secdiff * t;                                                               // rtrnmc :311-317
                ef = (double)(float)(1. - exp(-od));      // rounded as the array path stores it (cfef is float)
                any = 1;
            }
            const size_t o = ((size_t)(B - 1) * nlay + (lay - 1)) * ncb + col;
            W.odcld[o] = od;
            W.efcl[o] = ef;
            continue;
        }
#pragma unroll 1
        for (int qi = 0; qi < (ng + 3) / 4; qi++, quad++) {
            double cf[4], tau[4], ci[4], cw[4];
            if (FROMMASK) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int ig = g0 + 4 * qi + j;
                    const bool on = (4 * qi + j < ng) && ((mw[ig >> 5] >> (ig & 31)) & 1u);
                    cf[j] = on ? 1.0 : 0.0; tau[j] = on ? tband : 0.0; ci[j] = on ? mciwp : 0.0; cw[j] = on ? mclwp : 0.0;
                }
            } else {
                const size_t base = (size_t)(g0 + 4 * qi) + (size_t)NGPT * cl;     // 16-byte aligned: every band starts on an even g
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (4 * qi + 2 * h < ng) {              // band sizes are even: a pair is inside the band or outside it
                        const double2 a = *reinterpret_cast<const double2 *>(m.cldfmcl + base + 2 * h);
                        const double2 b = *reinterpret_cast<const double2 *>(m.taucmcl + base + 2 * h);
                        const double2 c = *reinterpret_cast<const double2 *>(m.ciwpmcl + base + 2 * h);
                        const double2 d = *reinterpret_cast<const double2 *>(m.clwpmcl + base + 2 * h);
                        cf[2 * h] = a.x; cf[2 * h + 1] = a.y; tau[2 * h] = b.x; tau[2 * h + 1] = b.y;
                        ci[2 * h] = c.x; ci[2 * h + 1] = c.y; cw[2 * h] = d.x; cw[2 * h + 1] = d.y;
                    } else {
                        cf[2 * h] = cf[2 * h + 1] = 0.0; tau[2 * h] = tau[2 * h + 1] = 0.0;
                        ci[2 * h] = ci[2 * h + 1] = 0.0; cw[2 * h] = cw[2 * h + 1] = 0.0;
                    }
                }
            }
            double od[4];
            float ocf[4], oef[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                double t = tau[j];
                if (cf[j] >= cldmin && (ci[j] + cw[j] >= cldmin || t >= cldmin)) {        // cldprmc :181-183
                    if (inflag == 1) err = E_MC_INFLAG1;                                     // :190-191
                    else if (inflag == 2) {
                        double a_i = 0.0, a_l = 0.0;
                        if (ci[j] != 0.0 && iceflag >= 0 && iceflag <= 3) { a_i = ai; if (ice_err) err = ice_err; }
                        if (cw[j] != 0.0) {
                            if (liqflag == 0) a_l = al;
                            else if (liqflag == 1) { a_l = al; if (liq_err) err = liq_err; }
                        }
                        t = ci[j] * a_i + cw[j] * a_l;                                          // :267-268
                    }
                }
                od[j] = 0.0; oef[j] = 0.f;
                if (cf[j] == 1.0) {                                                             // rtrnmc :311-317
                    od[j] = secdiff * t;
                    oef[j] = (float)((1. - exp(-od[j])) * cf[j]);
                    any = 1;
                }
                ocf[j] = (float)cf[j];
            }
            const size_t so = ((size_t)quad * nlay + (lay - 1)) * ncb + col;
            double2 *po = reinterpret_cast<double2 *>(W.odg + so * 4);
            po[0] = make_double2(od[0], od[1]);
            po[1] = make_double2(od[2], od[3]);
            float4 *pf = reinterpret_cast<float4 *>(W.cfef + so * 8);
            pf[0] = make_float4(ocf[0], ocf[1], ocf[2], ocf[3]);
            pf[1] = make_float4(oef[0], oef[1], oef[2], oef[3]);
        }
    }
    W.cflag[(size_t)lay * ncb + col] = any;
    if (any) atomicOr(&W.cflag[col], 8);
    if (any && lay > *reinterpret_cast<volatile int *>(&W.btop[col >> 6])) atomicMax(&W.btop[col >> 6], lay);
    if (any && lay < *reinterpret_cast<volatile int *>(&W.bbot[col >> 6])) atomicMin(&W.bbot[col >> 6], lay);
    if (lay == 1) W.cflag[(size_t)(nlay + 1) * ncb + col] = 0;
    if (err) atomicCAS(W.err, 0, err);
}

// ------------------------------------------------------------------------------------------------
// McICA sub-column generator: mcica_subcol_lw / generate_stochastic_clouds (src/mcica_subcol_gen_lw.f90:183-703).
// The GCM routine does not compile as shipped; the per-column semantics are those of its
// compilable one-column statement, src/mcica_subcol_gen_lw.1col.f90:171-710.
//   k_subcol_kiss  irng = 0: every column owns a KISS stream seeded from its four lowest layer pressures
//                  (:460-474) and consumed in (sub-column, layer) order -> a thread per (column, 8 sub-columns), each
//                  sub-column's stream reached by jump-ahead; the decisions leave as bytes of the bit mask.
//   k_subcol_slab  irng = 1: the Mersenne-Twister stream is ONE sequence over (sub-column, column, layer)
//                  (:497-503); k_mt_jump / k_mt_fill draw it chunk-parallel and this kernel applies the overlap rules to
//                  the slab of one sub-column.
//   k_subcol_expand  mask -> the (140,ncol,nlay) arrays of the reference interface (:664-680).
//   k_alpha        get_alpha (src/mcica_subcol_gen_lw.f90:68-180).
// Integer arithmetic is 32-bit wrap-around with logical shifts and the uniform deviates are formed without
// FMA contraction, so the cloud masks are bit-identical to the reference's.
// ------------------------------------------------------------------------------------------------
struct SubcolIn { const double *play, *cldfrac, *alpha; };      // each (ncol_total, nlay), device pointers

struct Kiss { unsigned a, b, c, d; };

// multiplier x (low half of s) + (high half of s): one v_mad_u32_u16, which reads the low halves of its factors (the compiler
// masks the half out first and multiplies with v_mad_u32_u24)
__device__ __forceinline__ unsigned kiss_mwc(unsigned s, unsigned mul)
{
    unsigned r;
    const unsigned carry = s >> 16;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(s), "s"(mul), "v"(carry));
    return r;
}
// 69069 a + 1327217885 (mod 2^32) from 24-bit multiplies, which run at full rate where the 32-bit v_mul_lo_u32 takes four issue
// slots: 69069 x (low 24 bits of a) + the constant, plus (0xCD = 69069 mod 256) x (top byte of a) shifted to the top byte
__device__ __forceinline__ unsigned kiss_lcg(unsigned a)
{
    unsigned h;
    const unsigned top = a >> 24;
    asm("v_mul_u32_u24_e32 %0, 0xcd, %1" : "=v"(h) : "v"(top));
    return (h << 24) + (__umul24(a, 69069u) + 1327217885u);
}
__device__ __forceinline__ void kiss_step(Kiss &s)             // kissvec's state advance, src/mcica_subcol_gen_lw.f90:711-745
{
    s.a = kiss_lcg(s.a);
    s.b ^= s.b << 13; s.b ^= s.b >> 17; s.b ^= s.b << 5;
    s.c = kiss_mwc(s.c, 18000u);
    s.d = kiss_mwc(s.d, 30903u);
}
__device__ __forceinline__ int kiss_int(Kiss &s)               // ... the integer it returns
{
    kiss_step(s);
    return (int)(s.a + s.b + (s.c << 16) + s.d);
}
__device__ __forceinline__ double kiss_unit(int k)             // ... and the deviate the reference forms from it
{
#pragma clang fp contract(off)
    const double r = (double)k * 2.328306e-10;
    return r + 0.5;
}
__device__ __forceinline__ double kiss_next(Kiss &s) { return kiss_unit(kiss_int(s)); }

// kiss_unit is strictly increasing in k (one step of k is 2.3e-10, an ulp of the sum at most 1.1e-16), so a comparison of a deviate
// with a value v that does not depend on other deviates is a comparison of the integers: kiss_unit(k) >= v  <=>  k >= T(v) with
// T(v) the smallest such k.  T = INT_MIN when every k qualifies, INT_MAX + 1 when none does (v = NaN included).
__device__ __forceinline__ long long kiss_threshold(double v)
{
#pragma clang fp contract(off)
    constexpr long long LO = -2147483648ll, HI = 2147483647ll;
    if (!(kiss_unit((int)HI) >= v)) return HI + 1;
    if (kiss_unit((int)LO) >= v) return LO;
    double g = (v - 0.5) / 2.328306e-10;                        // within a step or two of the answer, which lies in (LO, HI]
    g = g < -2147483647. ? -2147483647. : (g > 2147483647. ? 2147483647. : g);
    long long t = (long long)g;
    for (int it = 0; it < 64 && t > LO + 1 && kiss_unit((int)(t - 1)) >= v; it++) t--;
    for (int it = 0; it < 64 && t < HI && !(kiss_unit((int)t) >= v); it++) t++;
    return t;
}

// Jump-ahead of the stream by n draws.  The four component generators are each a linear map of their own state:
//   a  (congruential, mod 2^32)      a -> A a + B with (A, B) the n-fold composition;
//   b  (xorshift 13/17/5)            linear over GF(2): the images X[i] of the 32 unit vectors under n steps;
//   c, d (multiply-with-carry, multipliers 18000 / 30903, base 2^16)   the 32-bit word s = carry 2^16 + value satisfies
//        2^16 s' = s (mod m), m = multiplier 2^16 - 1, i.e. s' = multiplier s (mod m).  Two real steps bring any 32-bit word into
//        [0, m]; 0 and m are fixed points and every other word stays in [1, m-1], where the residue names the word.  So n >= 2 steps
//        are two steps and a multiplication by P = multiplier^(n-2) mod m.
// The host fills the constants (kissjump.hpp, kiss_jump_entry).
// (struct KissJump and the moduli KISS_MC / KISS_MD: kissjump.hpp)

template <unsigned M>
__device__ __forceinline__ unsigned kiss_mwc_mul(unsigned s, unsigned P)
{
    const unsigned r = (unsigned)(((unsigned long long)s * (unsigned long long)P) % (unsigned long long)M);
    return s == M ? M : r;
}
__device__ __forceinline__ void kiss_jump(Kiss &s, const KissJump &J)
{
    if (J.n < 2u) { if (J.n == 1u) kiss_step(s); return; }
    s.a = J.A * s.a + J.B;
    unsigned b = 0u;
#pragma unroll
    for (int i = 0; i < 32; i++) b ^= (0u - ((s.b >> i) & 1u)) & J.X[i];
    s.b = b;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        s.c = 18000u * (s.c & 65535u) + (s.c >> 16);
        s.d = 30903u * (s.d & 65535u) + (s.d >> 16);
    }
    s.c = kiss_mwc_mul<KISS_MC>(s.c, J.Pc);
    s.d = kiss_mwc_mul<KISS_MD>(s.d, J.Pd);
}

// overlap rule applied to the deviate of (sub-column, layer l) given the final deviate of layer l-1
__device__ __forceinline__ double overlap_rule(int icld, int l, double x, double x2, double prev, double cf_below, double alpha_l)
{
#pragma clang fp contract(off)
    if (l > 0) {
        if (icld == 2) {                                        // maximum-random, .1col :440-448
            const double one_m = 1. - cf_below;
            if (prev > one_m) x = prev;
            else x = x * one_m;
        } else if (icld == 4 || icld == 5) {                    // exponential(-random), .1col :492-496, :521-525
            if (x2 < alpha_l) x = prev;
        }
    }
    return x;
}

// k_subcol_kiss: a work-group is KJ_COLS columns x all sub-columns; a thread owns 8 consecutive sub-columns of one column (one byte
// of the mask) and walks the layers once, with the eight streams side by side: each was brought to its place in the column's stream
// by jump-ahead (the reference draws sub-column after sub-column, layer after layer - :475-530 - so sub-column s starts
// permuteseed + s x (draws per sub-column) draws into the stream).  What a decision is compared with depends on (column, layer)
// only: the work-group forms those values once, into LDS, as integer thresholds where the rule allows it (random, maximum,
// exponential: 1, 3, 4/5) and as 1 - cldfrac for the maximum-random rule, whose deviates are rescaled in floating point.
// Lanes 4 c .. 4 c + 3 of a wave hold the four bytes of one column's mask word and neighbouring columns follow, so one byte
// store per layer writes 64 contiguous bytes.  (The first version - a thread per column walking its 140 x nlay decisions in stream
// order with cldfrac / alpha re-read for every sub-column - moved 161 GB per 1e6 columns through HBM: 73 KB per wave of re-read
// working set, sixteen waves a CU, against 128 KB of L2 a CU.)
constexpr int KJ_COLS = 16;
constexpr int KJ_NWORD = MASK_WORDS;                 // mask words of a column = waves of a work-group
constexpr int KJ_BLOCK = 64 * KJ_NWORD;
constexpr int KJ_NGROUP = 4 * KJ_NWORD;             // 8-sub-column groups, the last ones possibly short or empty

// RULE 1 random, 2 maximum-random, 3 maximum, 4 exponential / exponential-random (icld 4 and 5 differ in alpha only).
// jt[g]: jump from the seed to the first sub-column of group g; jsub: jump by one sub-column.
// (waves per SIMD pinned to what the kernel needs without spilling: left to itself the allocator took 79 / 97 / 95 registers for rules
// 1 / 2 / 4 after an unrelated change of the argument struct - four waves instead of six, 3.7 -> 4.5 ms per 1e6 columns)
__host__ __device__ constexpr int kiss_waves(int rule) { return rule == 3 ? 5 : 6; }
template <int RULE>
__global__ __launch_bounds__(KJ_BLOCK, kiss_waves(RULE)) void k_subcol_kiss(Workspace W, SubcolIn in, const KissJump *jt, KissJump jsub, int ncol, int col0, int nb, int nlay)
{
#pragma clang fp contract(off)
    extern __shared__ int4 kj_thr[];                             // [nlay][KJ_COLS]
    __shared__ int kj_top;                                       // highest layer with cloud in any of the work-group's columns
    const int tid = threadIdx.x;
    if (tid == 0) kj_top = -1;
    __syncthreads();
    int top = -1;
    const int cl = (tid >> 2) & (KJ_COLS - 1);
    const int g = (tid >> 6) * 4 + (tid & 3);                    // sub-columns 8 g .. 8 g + 7 = byte g & 3 of mask word g >> 2
    const int cb = blockIdx.x * KJ_COLS;
    const double cldmin = 1.0e-20;
    // pass 1: the highest layer with cloud in any of the work-group's columns.  Every sub-column's stream is positioned on its own, so
    // nothing depends on the draws of the layers above it: no deviate reaches 1 - 0 there, the walk ends at `ltop` (the stratosphere
    // of every column, most of the troposphere of many) and thresholds are formed up to it only.
    for (int e = tid; e < nlay * KJ_COLS; e += KJ_BLOCK) {
        const int c = e & (KJ_COLS - 1), l = e / KJ_COLS;
        if (cb + c < nb && in.cldfrac[(size_t)col0 + (size_t)(cb + c) + (size_t)ncol * l] >= cldmin) top = l;       // (e ascends with l)
    }
    if (top >= 0) atomicMax(&kj_top, top);
    __syncthreads();
    const int ltop = kj_top;
    for (int e = tid; e < (ltop + 1) * KJ_COLS; e += KJ_BLOCK) {
        const int c = e & (KJ_COLS - 1), l = e / KJ_COLS;
        int4 t = make_int4(0, (int)0x80000000, 0, 0);
        if (cb + c < nb) {
            const size_t cell = (size_t)col0 + (size_t)(cb + c) + (size_t)ncol * l;
            double cf = in.cldfrac[cell];
            if (cf < cldmin) cf = 0.0;
            const double v = 1. - cf;                            // cloudy: deviate >= v (:655-661)
            if (RULE == 2) {
                t.x = __double2loint(v); t.y = __double2hiint(v);
            } else {
                const long long T = kiss_threshold(v);
                t.x = T > 2147483647ll ? 2147483647 : (int)T;
                t.w = T > 2147483647ll ? 0 : 0xff;               // no integer reaches v: the byte is cleared after the compare
                if (RULE == 4 && l > 0) {                        // deviate2 < alpha keeps the deviate of the layer below (:492-496)
                    const double al = in.alpha[cell];
                    const long long Ta = al != al ? -2147483648ll : kiss_threshold(al);
                    t.y = Ta > 2147483647ll ? 2147483647 : (int)Ta;
                    t.z = Ta > 2147483647ll ? 1 : 0;             // every integer is below alpha
                }
            }
        }
        kj_thr[e] = t;
    }
    __syncthreads();
    if (cb + cl >= nb) return;
    const size_t gc = (size_t)col0 + (size_t)(cb + cl);
    Kiss s[8];
    {
        const double p1 = in.play[gc] * 1.e2, p2 = in.play[gc + (size_t)ncol] * 1.e2;
        const double p3 = in.play[gc + (size_t)ncol * 2] * 1.e2, p4 = in.play[gc + (size_t)ncol * 3] * 1.e2;
        if (p1 < p2) { atomicCAS(W.err, 0, (int)E_KISS_PMID); return; }            // :463-466
        s[0].a = (unsigned)(int)((p1 - (double)(long long)p1) * 1000000000.);
        s[0].b = (unsigned)(int)((p2 - (double)(long long)p2) * 1000000000.);
        s[0].c = (unsigned)(int)((p3 - (double)(long long)p3) * 1000000000.);
        s[0].d = (unsigned)(int)((p4 - (double)(long long)p4) * 1000000000.);
    }
    // sub-columns past NGPT in the last word: their streams run along (no divergence in the loop below), their bits are cleared
    const int nv = min(8, max(0, NGPT - 8 * g));
    const unsigned live = (1u << nv) - 1u;
    unsigned char *out = reinterpret_cast<unsigned char *>(W.mask + ((size_t)(g >> 2) * nlay) * W.mask_stride + gc) + (g & 3);
    const size_t ostep = W.mask_stride * sizeof(unsigned);
    if (ltop < 0) {                                              // sixteen cloud-free columns
        for (int l = 0; l < nlay; l++) out[(size_t)l * ostep] = (unsigned char)0;
        return;
    }
    kiss_jump(s[0], jt[g]);
#pragma unroll
    for (int j = 1; j < 8; j++) { s[j] = s[j - 1]; kiss_jump(s[j], jsub); }
    int kx[8];
    double px[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { kx[j] = RULE == 3 ? kiss_int(s[j]) : 0; px[j] = 0.0; }
    double one_m = 1.0;
    int4 t = kj_thr[cl];
#pragma unroll 1
    for (int l = 0; l <= ltop; l++) {
        const int4 tn = kj_thr[min(l + 1, ltop) * KJ_COLS + cl];
        unsigned bits = 0u;
        if (RULE == 2) {                                         // maximum-random, .1col :440-448
            const double v = __hiloint2double(t.y, t.x);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                double x = kiss_next(s[j]);
                if (l > 0) x = px[j] > one_m ? px[j] : x * one_m;
                px[j] = x;
                bits |= x >= v ? 1u << j : 0u;
            }
            one_m = v;
        } else {
            const bool below = t.z != 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (RULE == 1) kx[j] = kiss_int(s[j]);
                if (RULE == 4) {
                    const int k = kiss_int(s[j]), k2 = kiss_int(s[j]);
                    kx[j] = ((k2 < t.y) | below) ? kx[j] : k;
                }
                bits |= kx[j] >= t.x ? 1u << j : 0u;
            }
            bits &= (unsigned)t.w;
        }
        out[(size_t)l * ostep] = (unsigned char)(bits & live);
        t = tn;
    }
    for (int l = ltop + 1; l < nlay; l++) out[(size_t)l * ostep] = (unsigned char)0;
}

// ---- Mersenne Twister on the device (irng = 1) ----------------------------------------------------------------------------------
// The stream is ONE MT19937 sequence over (sub-column, column, layer) (src/mcica_subcol_gen_lw.f90:497-503).  It is cut into chunks,
// the state at the start of every chunk is reached by jump-ahead (k_mt_jump applies g(x) = x^n mod phi(x) to a state by Horner's
// scheme; mtjump.hpp builds the polynomials and has the references), and k_mt_fill regenerates each chunk's deviates from its state.
// A state is 624 words in canonical order: st[0] = x_k (only its top bit matters) ... st[623] = x_{k+623}; deviate number t of the
// stream is the tempered x_{t+624}.
constexpr int MT_NW = 624, MT_BLOCK = 64, MT_PW = 312, MT_R = (MT_NW + 63) / 64;
// (One wavefront per state: with ten waves the two barriers around every XOR pass of k_mt_jump cost 11 ms per round of 19 937 steps and
// the four per block of k_mt_fill 1 us; a single wave needs no barrier beyond its own LDS ordering - 2 ms and 0.25 us.)

__device__ __forceinline__ unsigned mt_next(unsigned x0, unsigned x1, unsigned x397)
{
    const unsigned y = (x0 & 0x80000000u) | (x1 & 0x7fffffffu);
    return x397 ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// dst state = g(A) src state.  Work-group (x, y): src = st[src0 + y ystride + x xstride], dst = src + doff.
// Horner's scheme acc <- A acc ^ g_i s, taken 64 coefficients at a time (A is linear):
//      acc <- A^64 acc  ^  sum_{t=0..63} g_{i-t} A^(63-t) s.
// A^64 on a state is 64 new words, each a function of old words only (64 < 227): one word per lane.  A^m s is the window m .. m+623 of
// the source's own word sequence, so the sum is, for every word j, an XOR of ext[j + 63 - t] over the set coefficients - reads of a
// static LDS array, with the ring `acc` (logical word j at (h + j) mod 624) touched once per 64 coefficients.  (Coefficient by
// coefficient, with a barrier pair around every XOR pass, a round of jumps took 12 ms; this form 0.2 ms.)
__global__ __launch_bounds__(MT_BLOCK) void k_mt_jump(unsigned *st, const unsigned long long *g, int src0, int xstride, int ystride, int doff)
{
    __shared__ unsigned acc[MT_NW], ext[MT_NW + 64 + 16];
    __shared__ unsigned long long gs[MT_PW];
    const int tid = threadIdx.x;
    const size_t si = (size_t)src0 + (size_t)blockIdx.y * ystride + (size_t)blockIdx.x * xstride;
    const unsigned *sp = st + si * MT_NW;
    unsigned *dp = st + (si + (size_t)doff) * MT_NW;
#pragma unroll
    for (int r = 0; r < MT_R; r++) {
        const int j = tid + 64 * r;
        if (j < MT_NW) { ext[j] = sp[j]; acc[j] = 0u; }
    }
    for (int j = tid; j < MT_PW; j += 64) gs[j] = g[j];
    __syncthreads();
    ext[MT_NW + tid] = mt_next(ext[tid], ext[tid + 1], ext[tid + 397]);       // words 624 .. 687 of the source's sequence
    __syncthreads();
    // the XOR pass: lane `tid` owns the ten consecutive words 10 tid .. 10 tid + 9 (lane 62 four, lane 63 none).  Going through the 64
    // coefficients from offset 63 down, the window ext[j0 + b .. j0 + b + 9] slides by one word per coefficient: one LDS read each,
    // kept in a ring of ten registers whose slot numbers are compile-time constants in the unrolled loop.
    const int j0 = 10 * tid;
    const int nk = tid < 62 ? 10 : (tid == 62 ? 4 : 0);
    int h = 0;
    bool zero = true;                                                         // acc is still all zero (leading zero coefficients)
    for (int w = MT_PW - 1; w >= 0; w--) {
        const unsigned long long mv = gs[w];                                  // coefficients 64 w + 63 (offset 63) ... 64 w (offset 0)
        const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)mv), mhi = __builtin_amdgcn_readfirstlane((unsigned)(mv >> 32));
        if (zero && (mlo | mhi) == 0u) continue;
        if (!zero) {                                                          // acc <- A^64 acc
            const int p0 = h + tid < MT_NW ? h + tid : h + tid - MT_NW;
            const int p1 = p0 + 1 < MT_NW ? p0 + 1 : 0;
            const int p397 = p0 + 397 < MT_NW ? p0 + 397 : p0 + 397 - MT_NW;
            const unsigned nw = mt_next(acc[p0], acc[p1], acc[p397]);
            __syncthreads();
            acc[p0] = nw;                                                     // the 64 oldest words make room for the 64 newest
            h = h + 64 < MT_NW ? h + 64 : h + 64 - MT_NW;
            __syncthreads();
        }
        if ((mlo | mhi) != 0u) {
            unsigned v[10], P[10];
#pragma unroll
            for (int k = 0; k < 10; k++) { v[k] = 0u; P[(63 + k) % 10] = ext[j0 + 63 + k]; }
#pragma unroll
            for (int b = 63; b >= 0; b--) {
                const bool on = b >= 32 ? ((mhi >> (b - 32)) & 1u) != 0u : ((mlo >> b) & 1u) != 0u;       // (uniform)
                if (on) {
#pragma unroll
                    for (int k = 0; k < 10; k++) v[k] ^= P[(b + k) % 10];
                }
                if (b > 0) P[(b - 1) % 10] = ext[j0 + b - 1];                 // replaces word j0 + b + 9, which has the same slot
            }
#pragma unroll
            for (int k = 0; k < 10; k++) {
                if (k < nk) { const int j = j0 + k; const int p = h + j < MT_NW ? h + j : h + j - MT_NW; acc[p] ^= v[k]; }
            }
            zero = false;
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < MT_R; r++) {
        const int j = tid + 64 * r;
        if (j < MT_NW) { const int p = h + j < MT_NW ? h + j : h + j - MT_NW; dp[j] = acc[p]; }
    }
}

// Deviates of chunk blockIdx.x of slab blockIdx.y: rnd[y per + m C .. y per + min((m+1) C, per)) from the state st[state0 + y M + m].  A block of 624 new words
// depends on the previous 624 in three waves of the recurrence (i < 227 on old words only, 227 <= i < 454 on the first wave's words,
// the rest on the second's; word 623 needs the new word 0).  genrand_real1: tempered word / (2^32 - 1) (src/mcica_random_numbers.f90:262-295).
__global__ __launch_bounds__(MT_BLOCK) void k_mt_fill(const unsigned *st, double *rnd, int state0, int M, unsigned long long C, unsigned long long per)
{
#pragma clang fp contract(off)
    __shared__ unsigned a[MT_NW], b[MT_NW];
    const int tid = threadIdx.x;
    const unsigned long long lo = (unsigned long long)blockIdx.x * C;
    if (lo >= per) return;
    const unsigned long long len = (per - lo < C) ? per - lo : C;
    rnd += (size_t)blockIdx.y * per;                             // slab blockIdx.y of this launch: states state0 + y M ...
    for (int j = tid; j < MT_NW; j += 64) a[j] = st[((size_t)state0 + (size_t)blockIdx.y * M + blockIdx.x) * MT_NW + j];
    __syncthreads();
    unsigned *o = a, *n = b;
    for (unsigned long long base = 0; base < len; base += MT_NW) {
#pragma unroll
        for (int r = 0; r < 4; r++) { const int i = tid + 64 * r; if (i < 227) n[i] = mt_next(o[i], o[i + 1], o[i + 397]); }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) { const int i = 227 + tid + 64 * r; if (i < 454) n[i] = mt_next(o[i], o[i + 1], n[i - 227]); }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; r++) { const int i = 454 + tid + 64 * r; if (i < 623) n[i] = mt_next(o[i], o[i + 1], n[i - 227]); }
        if (tid == 0) n[623] = mt_next(o[623], n[0], n[396]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MT_R; r++) {
            const int i = tid + 64 * r;
            if (i < MT_NW && base + i < len) {
                unsigned y = n[i];
                y ^= y >> 11;
                y ^= (y << 7) & 0x9d2c5680u;
                y ^= (y << 15) & 0xefc60000u;
                y ^= y >> 18;
                rnd[lo + base + i] = (double)y / 4294967295.0;
            }
        }
        unsigned *t = o; o = n; n = t;
    }
}

// rnd: the deviates of sub-column `isub` for all columns in stream order: icld == 3: [ncol]; otherwise
// [ncol][nlay][nd] with nd = 2 for icld 4/5 (CDF, CDF2), else 1.  The mask must have been zeroed.
constexpr int SLAB_GROUP = 8;       // slabs a thread takes through a chunk of layers on one load of cldfrac / alpha
__global__ __launch_bounds__(256) void k_subcol_slab(Workspace W, SubcolIn in, const double *rnd, int ncol, int nlay, int icld, int isub0, int nslab,
                                                     unsigned long long per)
{
#pragma clang fp contract(off)
    const size_t gc = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gc >= (size_t)ncol) return;
    // several slabs per launch: the bits of a mask word come from different work-groups, hence the atomic OR
    const int s0 = blockIdx.y * SLAB_GROUP, ns = min(SLAB_GROUP, nslab - s0);
    rnd += (size_t)s0 * per;
    const double cldmin = 1.0e-20;
    const bool two = icld == 4 || icld == 5;
    const int nd = two ? 2 : 1;
    double prev[SLAB_GROUP], r3[SLAB_GROUP];
#pragma unroll
    for (int j = 0; j < SLAB_GROUP; j++) { prev[j] = 0.0; r3[j] = (icld == 3 && j < ns) ? rnd[(size_t)j * per + gc] : 0.0; }
    double cf_below = 0.0;
    // layers in chunks of 8: cloud fractions and overlap parameters are requested once for the group's slabs, each slab's deviates of the
    // chunk (contiguous per column in stream order) together, then the eight dependent steps run from registers
#pragma unroll 1
    for (int l0 = 0; l0 < nlay; l0 += 8) {
        double cfc[8], alc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int l = l0 + i < nlay ? l0 + i : nlay - 1;
            double cf = in.cldfrac[gc + (size_t)ncol * l];
            cfc[i] = cf < cldmin ? 0.0 : cf;
            alc[i] = two ? in.alpha[gc + (size_t)ncol * l] : 0.0;
        }
#pragma unroll 1
        for (int j = 0; j < ns; j++) {
            const double *rj = rnd + (size_t)j * per;
            double xs[8], x2s[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int l = l0 + i < nlay ? l0 + i : nlay - 1;
                const size_t o = (gc * nlay + l) * nd;
                xs[i] = icld == 3 ? 0.0 : rj[o];
                x2s[i] = two ? rj[o + 1] : 0.0;
            }
            const int isub = isub0 + s0 + j;
            const int w = isub >> 5;
            const unsigned bit = 1u << (isub & 31);
            double pv = 0.0, r3j = 0.0;
#pragma unroll
            for (int jj = 0; jj < SLAB_GROUP; jj++) if (jj == j) { pv = prev[jj]; r3j = r3[jj]; }
            double below = cf_below;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int l = l0 + i;
                if (l < nlay) {
                    const double x = overlap_rule(icld, l, icld == 3 ? r3j : xs[i], x2s[i], pv, below, alc[i]);
                    pv = x; below = cfc[i];
                    if (x >= 1. - cfc[i]) atomicOr(&W.mask[((size_t)w * nlay + l) * W.mask_stride + gc], bit);
                }
            }
#pragma unroll
            for (int jj = 0; jj < SLAB_GROUP; jj++) if (jj == j) prev[jj] = pv;
        }
        cf_below = cfc[7];                                    // (a short last chunk ends the loop)
    }
}

__device__ __forceinline__ int g_band(int ig)            // 1-based band of 0-based g-point ig (ngb)
{
    int b = 1;
#pragma unroll
    for (int B = 2; B <= NBND; B++) b += (ig >= band_g0(B)) ? 1 : 0;
    return b;
}

// one thread per (sub-column, column) of one layer: fully coalesced stores into the (140,ncol,nlay) arrays.
// Layers l0 .. l0 + gridDim.y - 1 are written to the output arrays starting at their layer 0.
__global__ __launch_bounds__(256) void k_subcol_expand(Workspace W, const double *ciwp, const double *clwp, const double *tauc,
                                                       int ncol, int nlay, int l0, double *cldfmcl, double *ciwpmcl, double *clwpmcl, double *taucmcl)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)NGPT * ncol) return;
    const int l = l0 + blockIdx.y;
    const int isub = (int)(e % NGPT);
    const size_t gc = e / NGPT;
    const unsigned word = W.mask[((size_t)(isub >> 5) * nlay + l) * W.mask_stride + gc];
    const bool on = (word >> (isub & 31)) & 1u;
    const size_t cl = gc + (size_t)ncol * l;
    const size_t o = e + (size_t)NGPT * ncol * blockIdx.y;
    cldfmcl[o] = on ? 1.0 : 0.0;
    ciwpmcl[o] = on ? ciwp[cl] : 0.0;
    clwpmcl[o] = on ? clwp[cl] : 0.0;
    taucmcl[o] = on ? tauc[(g_band(isub) - 1) + (size_t)NBND * cl] : 0.0;
}

__global__ __launch_bounds__(256) void k_alpha(int ncol, int nlay, int icld, int idcor, double decorr_con, const double *dz,
                                               const double *lat, int juldat, const double *cldfrac, double *alpha)
{
    const size_t gc = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gc >= (size_t)ncol) return;
    const int k = blockIdx.y;                       // layer index 0 .. nlay-1
    if (!(icld == 4 || icld == 5)) return;
    const double am1 = 1.4315, am2 = 2.1219, am4 = -25.584, amr = 7.0;
    double decorr_len, decorr_inv = 1.0;
    if (idcor == 1) {                               // latitude / day-of-year dependent decorrelation length, :140-152
        double am3;
        if (juldat > 181) am3 = -4. * amr / 365. * (juldat - 272);
        else am3 = 4. * amr / 365. * (juldat - 91);
        const double y = lat[gc] - am3;
        decorr_len = (am1 + am2 * exp(-(y * y) / (am4 * am4))) * 1.e3;
    } else {
        decorr_len = decorr_con;
    }
    if (decorr_len >= 0.0) decorr_inv = 1.0 / decorr_len;
    const size_t o = gc + (size_t)ncol * k;
    double a = 0.0;
    if (k > 0) {
        const size_t om = o - (size_t)ncol;
        a = exp(-(0.5 * (dz[o] + dz[om])) * decorr_inv);
        if (icld == 5 && cldfrac[o] == 0.0 && cldfrac[om] > 0.0) a = 0.0;
    }
    alpha[o] = a;
}

// ------------------------------------------------------------------------------------------------
// The vertical recurrences: k_sweepc (cloud-free calls and the layers above the batch's highest cloud) and k_sweepz (the cloud zone:
// rtrn, rtrnmr, rtrnmc), below.  Per cell they read k_layer's 4-byte code (cell_code) and form, in float64: transmittance and tfn
// factor (series, or the transmittance table staged in LDS; rtrn :372-451), the layer's Planck functions for the band (setcoef
// :173-269, from tlay / tlev and the band's totplnk row staged in LDS), the Planck fractions (taumol: constant per g-point, or
// interpolated between two fracref rows staged in LDS with the layer's (js, fs) from k_layer) and from them the source terms
// (rtrn :447-449).  The flux contributions (sum over g-points x 0.5 x delwave, rtrn :549-562) are added over a group of bands in LDS
// and written once per group and level.  Lanes beyond the last column work on a copy of the last column; only their stores are masked.
// ------------------------------------------------------------------------------------------------
struct SweepArgs {
    unsigned long long bands;  // the group's bands (all with the same number of quads), one nibble (band - 1) each
    int nbands, ncb;           // number of those bands, number of column blocks (= workgroups)
    int group;                 // index of the group's partial slabs (W.gdn1 ..); split: of its first band's
    int split;                 // 1: a batch that does not fill the chip - grid.y = the group's bands, ONE band per workgroup (a quarter of the
                               // waves per CU, four times the CUs: a level's latency is the dependent issue of one wave, not of three), a slab
                               // per band; k_flux adds the bands of a group first, in the list's order - what the group's workgroup does in LDS
                               // otherwise - so the fluxes do not depend on it
    int ncol, col0, nct, idrv;
    int istart, iend;          // only bands in [istart, iend] are swept
    const double *emis;        // semiss (nct,16)
    const double *cldfrac;     // (nct,nlay)
    const double *tlay;        // (nct,nlay)     layer temperatures    (tlay | tavel)
    const double *tlev;        // (nct,nlay+1)   interface temperatures (tlev | tz)
};

// LDS of a sweep workgroup: transmittance table (float pairs), per band the Planck rows and the fraction rows
#ifdef RRLW_SWEEP_EXPF
constexpr int SWEEP_LUT_BYTES = 16;                   // no table in LDS
#else
constexpr int SWEEP_LUT_BYTES = 8 * (NTBL + 1) + 8;
#endif
constexpr int SWEEP_PL_BYTES = 2 * 184 * 8, SWEEP_FR_BYTES = 16 * 16 * 8;
// Loads of the sweeps: wave-uniform base pointer in a buffer descriptor (scalar registers), per-lane 32-bit byte offset that does not
// change from level to level - no 64-bit vector address arithmetic per load.  nt = streaming (non-temporal) access.
// soff (round 5): the ROW of the level as a wave-uniform 32-bit byte offset in the instruction's scalar-offset operand.  Until then every
// load of a level formed its row's 64-bit address (clamp, a signed 64-bit multiply by the row stride, the add, the descriptor's second
// word: ~14 scalar instructions, ~106 per level and thread in k_sweepz), and the scalar stream is NOT hidden behind the vector one at
// three waves per SIMD: 100 scalar instructions more per level cost the sweeps 10 % (knock-in, profiles/round5_instruction_diet.md).
// Now the descriptor is built from the array's start (the compiler keeps it across the level loop) and a level costs the workspace arrays
// one multiply for the row and a shift per element size; the rows of the caller's arrays, whose column stride is the whole call's column
// count, keep a 64-bit address, formed from an unsigned 32 x 32 -> 64 bit product (row_ptr).  The driver keeps every row offset below
// 2^32 bytes (driver.hip: eff_batch).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sweep_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7ffffff0, 0x00020000);
}
// a quad record of cell codes (voff = the lane's column x CODE_BYTES), streaming
__device__ __forceinline__ pk4 bload_pk4_nt(const void *base, unsigned voff, unsigned soff = 0u)
{
    pk4 r;
    if constexpr (CODE_WORDS == 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(sweep_rsrc(base), (int)voff, (int)soff, 2);
        __builtin_memcpy(&r, &v, 16);
    } else if constexpr (CODE_WORDS == 3) {
        typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(sweep_rsrc(base), (int)voff, (int)soff, 2);
        r.w[0] = v[0]; r.w[1] = v[1]; r.w[2] = v[2];
    } else {
        typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
        const u32x2_ v = __builtin_amdgcn_raw_buffer_load_b64(sweep_rsrc(base), (int)voff, (int)soff, 2);
        r.w[0] = v[0]; r.w[1] = v[1];
    }
    return r;
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double bload_f64(const void *base, unsigned voff, unsigned soff = 0u)
{
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(sweep_rsrc(base), (int)voff, (int)soff, 0);
    double r;
    __builtin_memcpy(&r, &v, 8);
    return r;
}
__device__ __forceinline__ double2 bload_f64x2(const void *base, unsigned voff, unsigned soff = 0u)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(sweep_rsrc(base), (int)voff, (int)soff, 0);
    double2 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
__device__ __forceinline__ unsigned bload_u32(const void *base, unsigned voff, unsigned soff = 0u)
{
    return __builtin_amdgcn_raw_buffer_load_b32(sweep_rsrc(base), (int)voff, (int)soff, 0);
}
// row `r` of a caller's (column-fastest) array whose rows lie `stride` doubles apart: a 64-bit address from an unsigned 32 x 32 product
__device__ __forceinline__ const double *row_ptr(const double *base, unsigned r, unsigned stride)
{
    return base + (unsigned long long)r * (unsigned long long)stride;
}

__device__ __forceinline__ void bstore_f64(void *base, unsigned voff, double v)
{
    u32x2 x;
    __builtin_memcpy(&x, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(x, sweep_rsrc(base), (int)voff, 0, 0);
}
__device__ __forceinline__ void bstore_f64x2(void *base, unsigned voff, double v0, double v1)
{
    const double t[2] = {v0, v1};
    u32x4 x;
    __builtin_memcpy(&x, t, 16);
    __builtin_amdgcn_raw_buffer_store_b128(x, sweep_rsrc(base), (int)voff, 0, 0);
}

// Knock-IN (measurement, -DRRLW_KI_SALU=n): n more scalar instructions per level and thread in the sweeps, a dependent chain like the
// address arithmetic of the level's loads - what the scalar stream costs is what adding to it costs (profiles/round5_instruction_diet.md).
__device__ __forceinline__ void ki_salu(int &x)
{
#ifdef RRLW_KI_SALU
#pragma unroll
    for (int i = 0; i < RRLW_KI_SALU; i++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(x));
#else
    (void)x;
#endif
}

// table index of a cell's code (0 for the series branch: entry 0 is {1 - exp = 0, tfn = 0})
// From here to the end of k_sweepz nothing is contracted behind the source's back: every fused multiply-add of the sweeps is written as
// one (fma).  The two sweep kernels, and the copies of a level's body inside each (one per prefetch slot), must round a level alike -
// which of them sweeps a level follows from the clouds of the 64-column block a column shares - and with contraction left to the
// compiler they did not: tp + f * (tq1 - tq0) of planck_at fused in one inlined copy and not in another moved a flux partial by an ulp
// (tools/soak_host_entry.py, tests/test_hip_parity.py::test_a_column_does_not_depend_on_its_neighbours).
#pragma clang fp contract(off)
__device__ __forceinline__ unsigned code_index(scr_t c) { return (unsigned)max((int)(-c), 0); }
// the optical depth of a series cell, 0 for a table cell: max(c, 0) on the bit pattern (a negative float is a negative integer)
__device__ __forceinline__ float code_od(scr_t c) { return __int_as_float(max(__float_as_int(c), 0)); }
// (1 - transmittance, tfn factor) of a cell from its code and the table entry at code_index: rtrn :439-451.  Branch-free: a table
// cell adds the series terms of od = 0, a series cell adds the table terms of entry 0 - both exact zeros.
__device__ __forceinline__ void decode(scr_t c, const float2 &e, double &atr, double &tfn)
{
#ifdef RRLW_DECODE_F64
    const double od = (double)code_od(c);
    atr = (double)e.x + fma(-0.5 * od, od, od);
    tfn = fma(0.166667, od, (double)e.y);
#else
    // float arithmetic up to the two conversions: the table values are floats already, and of the two summands one is an exact
    // zero; the series terms carry ~1e-7 relative error (od <= 0.06), the same class as the float table entries
    const float od = code_od(c);
    atr = (double)(e.x + __builtin_fmaf(-0.5f * od, od, od));
    tfn = (double)__builtin_fmaf(0.166667f, od, e.y);
#endif
}
// RRLW_SWEEP_EXPF (measurement variant, north_star "__builtin_amdgcn_expf for transmittance"): the table entry of index i evaluated
// instead of looked up - the same closed forms rrtmg_lw_ini tabulates (src/rrtmg_lw_init.f90:125-142: t = i / 1e4,
// tau = bpade t / (1 - t), 1 - exp(-tau), tfn = tau / 6 below 0.06 else 1 - 2 (1 / tau - exp / (1 - exp))) in float32 with the hardware
// exp2 / rcp.  The INDEX is still the reference's (formed in float64 by k_layer).  Frees the table's 80 KB of LDS; costs ~12 more
// instructions per cell (4 of them transcendental) and float32 cancellation in 1 - exp and in tfn.  Measured: profiles/round2_expf.md.
__device__ __forceinline__ float2 lut_entry_expf(unsigned idx, float bpade)
{
    const float t = (float)idx * 1.0e-4f;
    const float tau = bpade * t * __builtin_amdgcn_rcpf(1.0f - t);                    // idx = 10000 -> +inf -> exp = 0, entry {1, 1}
    const float ex = __builtin_amdgcn_exp2f(-1.44269504088896f * tau);
    const float atr = 1.0f - ex;
    float tfn = tau < 0.06f ? tau * (1.0f / 6.0f) : 1.0f - 2.0f * (__builtin_amdgcn_rcpf(tau) - ex * __builtin_amdgcn_rcpf(atr));
    if (idx == 0u) tfn = 0.0f;                                                         // entry 0 = {0, 0} (series cells)
    if (idx >= 10000u) tfn = 1.0f;
    return make_float2(idx == 0u ? 0.0f : atr, tfn);
}
#ifdef RRLW_SWEEP_EXPF
#define RRLW_LUT_ENTRY(lut, idx) lut_entry_expf((idx), 3.5971223f)
#else
#define RRLW_LUT_ENTRY(lut, idx) (lut)[(idx)]
#endif

// integrated Planck function at temperature t from one band's row of totplnk: setcoef :173-269
__device__ __forceinline__ double planck_at(const double *tp, const double *tq, double t)
{
    const double x = t - 159.;
    const int ind = clampi((int)x, 1, 180);
    const double f = x - (double)ind;
    return fma(f, tq[ind] - tq[ind - 1], tp[ind - 1]);
}

// LDS tables of a sweep workgroup (one band): the transmittance table as float pairs {1 - exp, tfn}, the band's row of totplnk (and band
// 16's, for the istart = 16 variant of setcoef :233-246), its Planck-fraction rows (0-8 fracrefa, 9-13 fracrefb, 14-15 zeros).  Ends
// with a barrier.
__device__ __forceinline__ void sweep_stage_lut(const DevTables &T, unsigned char *smem, int tid, int nth)
{
    const double2 *src = reinterpret_cast<const double2 *>(T.stat + T.sl.lutf);
    double2 *dst = reinterpret_cast<double2 *>(smem);
    // (loads first, then the LDS writes, eight at a time: a plain copy loop waits for every load before it issues the next)
#ifdef RRLW_SWEEP_EXPF
    constexpr int NLUT = 0;
#else
    constexpr int NLUT = SWEEP_LUT_BYTES / 16;
#endif
    for (int i0 = tid; i0 < NLUT; i0 += 8 * nth) {
        double2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = src[min(i0 + k * nth, NLUT - 1)];
#pragma unroll
        for (int k = 0; k < 8; k++) if (i0 + k * nth < NLUT) dst[i0 + k * nth] = v[k];
    }
}
// one band's Planck integrals and Planck-fraction rows
__device__ __forceinline__ void sweep_stage_band(const DevTables &T, double (*s_pl)[184], double (*s_fr)[16], int B, bool alt16, bool lo_bin, bool up_bin,
                                                 int tid, int nth)
{
    const double *tp = alt16 ? T.stat + T.sl.totplk16 : T.stat + T.sl.totplnk + 181 * (B - 1);
    const double *tq = T.stat + T.sl.totplnk + 181 * 15;
    for (int i = tid; i < 181; i += nth) { s_pl[0][i] = tp[i]; s_pl[1][i] = tq[i]; }
    const int ng = T.band[B - 1].ng, fa = T.band[B - 1].fracrefa, fb = T.band[B - 1].fracrefb;
    const int na = lo_bin ? 9 : 1, nb = up_bin ? 5 : 1;
    for (int i = tid; i < 16 * 16; i += nth) {
        const int r = i >> 4, g = i & 15;
        double v = 0.0;
        if (g < ng) {
            if (r < 9) { if (r < na) v = T.ktab[fa + r * ng + g]; }
            else if (r < 14 && fb >= 0 && r - 9 < nb) v = T.ktab[fb + (r - 9) * ng + g];
        }
        s_fr[r][g] = v;
    }
}
// ------------------------------------------------------------------------------------------------
// k_sweepc : the clear-sky recurrences (src/rrtmg_lw_rtrn.f90:437-466 downward, :497-540 upward; identical in rtrnmr / rtrnmc) with
//            ALL g-points of a band in one thread: one thread = (column, band), no reduction across threads.  The work of a level that
//            does not depend on the g-point (two Planck interpolations, fraction-row selection, the loads of temperatures and codes,
//            loop control) is done once for up to 16 g-points instead of once per quad, and the wave has up to 16 independent
//            recurrences in flight.
//   PHASE 0  the whole column of a cloud-free call (icld = 0): downward, surface (rtrn :476-495), upward
//   PHASE 1  layers ltop+1 .. nlay downward, for the cloudy modes: above its block's hand-off level ltop (W.hgrp, k_blocksort) every column
//            of a wave is clear and the clear-sky stream equals the total one; the radiances at level ltop go to k_sweepz through W.hand
//   PHASE 2  layers ltop+1 .. nlay upward, total and clear-sky streams, starting from the radiances k_sweepz left in W.hand
// ------------------------------------------------------------------------------------------------
// waves per SIMD each instantiation is compiled for = what its registers allow without spilling (a spilled register is reloaded with a
// scratch load, which counts as a vector-memory operation: its wait drains every prefetch in flight), at most 4 (one workgroup of
// <= 1024 threads per CU: the transmittance table takes half of the LDS).  Register needs, unconstrained, by quads per thread 4 / 3 / 2 / 1:
//   phase 1                       154 148 102  84          (as compiled since the fused multiply-adds are written out: 136 114  91  75)
//   phase 0                       247 171 147 106          (two waves per band at 4 quads: 112; 140 111  85)
//   phase 2                       256+ 222 172 102         (145 with two waves per band; 188 143  95)
//   phase 0 with d/dT             250 199 148 105          (146; 189 144  95)
//   phase 2 with d/dT             256+ 256+ 241 132        (212; 256 + 44 accumulation registers; 210 126)
#ifndef RRLW_SWEEPC_P0
#define RRLW_SWEEPC_P0 6, 4, 3, 2      // (6, 3, 2, 2 until the sweeps' fused multiply-adds were written out: 111 / 140 registers at 2 / 3 quads since; 1e4 clear columns 0.78 -> 0.69 ms)
#endif
#ifndef RRLW_SWEEPC_P2D
#define RRLW_SWEEPC_P2D 3, 2, 1, 1
#endif
#ifndef RRLW_SWEEPC_WAVES_CAP
#define RRLW_SWEEPC_WAVES_CAP 4
#endif
__host__ __device__ constexpr int sweepc_waves(int NQ, int PHASE, bool IDRV)
{
    const int row = PHASE == 1 ? 0 : (PHASE == 0 && !IDRV) ? 1 : (PHASE == 2 && IDRV) ? 3 : 2;
    const int tab[4][4] = {{7, 5, 4, 3}, {RRLW_SWEEPC_P0}, {5, 3, 2, 1}, {RRLW_SWEEPC_P2D}};      // [row][NQ - 1]
    const int w = tab[row][NQ - 1];
    return w < RRLW_SWEEPC_WAVES_CAP ? w : RRLW_SWEEPC_WAVES_CAP;
}
// A workgroup = the bands of one GROUP (threadIdx.y; all with NQ quads) x column sub-blocks of 64 (threadIdx.z), one workgroup per CU
// (the transmittance table takes half of the LDS).  The per-level partials of the group's bands are added in LDS, in the order of the
// group's band list, every RRLW_SWEEPC_CODES levels: one 8- or 16-byte partial per group, level and column goes to HBM instead of one
// per band (16 bands -> 4 groups: the partial slabs were a quarter of this kernel's traffic and all of k_flux's).
constexpr int NGROUP_MAX = 8;
constexpr int SWEEPC_BAND_BYTES = SWEEP_PL_BYTES + SWEEP_FR_BYTES;
// values per band, level and column in the reduction buffer: total [, clear] [, their d/dT]
__host__ __device__ constexpr int sweepc_nval(int PHASE, bool IDRV) { return (PHASE == 2 ? 2 : 1) * ((IDRV && PHASE != 1) ? 2 : 1); }
#ifndef RRLW_SWEEPC_QUAD_BARRIER
#define RRLW_SWEEPC_QUAD_BARRIER 0   // 1: keep the quads of a level apart in the instruction schedule (fewer registers, less overlap; measured slower)
#endif
#ifndef RRLW_SWEEPC_CODES
#define RRLW_SWEEPC_CODES 2       // code slots: levels of cell codes in flight (the codes are the one HBM stream of the sweep; 4 registers per quad and slot)
#endif
constexpr int SWEEPC_LDS_MAX = 160 * 1024;
__host__ __device__ constexpr int sweepc_lds_bytes(int PHASE, bool IDRV, int nb, int nsb, int NT = 1)
{
    return SWEEP_LUT_BYTES + nb * SWEEPC_BAND_BYTES + 2 * sweepc_nval(PHASE, IDRV) * nb * NT * RRLW_SWEEPC_CODES * nsb * 64 * 8;
}
// threads per band: two where a band's 16 g-points with two streams would otherwise leave one wave per SIMD
#ifndef RRLW_SWEEPC_SPLIT
#define RRLW_SWEEPC_SPLIT 1
#endif
#ifndef RRLW_SWEEPC_SPLIT3
#define RRLW_SWEEPC_SPLIT3 0
#endif
__host__ __device__ constexpr int sweepc_nt(int NQ, int PHASE, bool IDRV)
{
    // 12 g-points x {total, clear} x {radiance, d/dT} in one thread need 256 registers + 70 accumulation registers as spill space: one wave
    // per SIMD.  Three threads of one quad (RRLW_SWEEPC_SPLIT3 = 1: three waves per SIMD, nothing spilled) repeat the level's Planck terms,
    // fraction rows and temperature loads three times: 8.26 ms against 5.59 per 5e5 137-layer columns - the one-wave form stays.
    if (RRLW_SWEEPC_SPLIT3 && NQ == 3 && PHASE == 2 && IDRV) return 3;
#ifdef RRLW_SWEEPC_SPLIT_P1       // measurement: the downward clear-sky sweep of the 16-g-point bands on two threads as well
    if (RRLW_SWEEPC_SPLIT && NQ == 4) return 2;
#endif
    return (RRLW_SWEEPC_SPLIT && NQ == 4 && (PHASE == 2 || PHASE == 0)) ? 2 : 1;
}
// bands per group: what fits the wave slots of the most register-hungry instantiation (phase 2 with d/dT)
__host__ __device__ constexpr int sweepc_group_cap(int NQ)
{
    int cap = 5;
    for (int ph = 0; ph < 3; ph++)
        for (int i = 0; i < 2; i++) {
            if (ph == 1 && i == 1) continue;
            const bool idrv = i == 1;
            const int nt = sweepc_nt(NQ, ph, idrv), w = 4 * sweepc_waves(NQ / nt, ph, idrv) / nt;      // wave slots
            if (w < cap) cap = w;
            while (cap > 1 && sweepc_lds_bytes(ph, idrv, cap, 1, nt) > SWEEPC_LDS_MAX) cap--;           // reduction buffer
        }
    return cap;
}
// column sub-blocks per workgroup for a group of nb bands
__host__ __device__ constexpr int sweepc_nsb(int NQ, int PHASE, bool IDRV, int nb)
{
    const int nt = sweepc_nt(NQ, PHASE, IDRV);
    int nsb = nsb_fit(4 * sweepc_waves(NQ / nt, PHASE, IDRV) / (nb * nt));         // (a divisor of SORT_GROUP: workgroups never straddle hand-off groups)
    while (nsb > 1 && sweepc_lds_bytes(PHASE, IDRV, nb, nsb, nt) > SWEEPC_LDS_MAX) nsb = nsb_fit(nsb - 1);
    return nsb;
}

// One slot of the rolling prefetch.  Every load is issued on every path (rows clamped to 1 .. nlay instead of skipped): the compiler can
// then count the loads in flight behind the one it needs and wait with vmcnt(N) instead of vmcnt(0) - vector-memory operations complete
// in order, so a wait for everything would stall the wave on the codes it has just requested for the next level.
template <int G> struct SweepcLev { double tl, tz; unsigned w; };

// 1 - x as a value of its own.  With contraction allowed across statements the compiler folds (1 - x) * y into fma(-x, y, y) where the
// difference has a single use and keeps the product where it has several: the same source line then rounds differently in two copies of
// the level body (the two prefetch slots, the two kernels) and a level's d(flux)/dT depended, in its last bit, on the hand-off level of
// its block - found by tools/soak_host_entry.py.  The empty asm hides the subtraction from that fold.
__device__ __forceinline__ double one_minus(double x)
{
    double t = 1.0 - x;
    asm("" : "+v"(t));
    return t;
}
// ... and a product as a value of its own: d(radiance)/dT of a level is the product radiance' x (1 - a), which is both the state carried
// upward and a term of the level's sum over the g-points - fused into that sum (fma(radiance', 1 - a, next term)) in one copy of the body
// and not in another, the sum rounded differently.  (The radiances themselves are results of an fma; nothing is left to fuse.)
__device__ __forceinline__ double times(double x, double y)
{
    double t = x * y;
    asm("" : "+v"(t));
    return t;
}

// NT threads (waves) per band, each with G = NQ / NT of its quads: the band's partial is the sum of their raw quad sums, formed - with the
// band's weight - by the wave that adds the group (so a band split over two waves still rounds like one swept by a single thread).  Used
// where all 16 g-points of a band with two streams would leave one wave per SIMD (sweepc_nt).
template <int NQ, int PHASE, bool IDRV, int NT = 1>
__global__ __launch_bounds__(256 * sweepc_waves(NQ / NT, PHASE, IDRV), sweepc_waves(NQ / NT, PHASE, IDRV)) void k_sweepc(DevTables T, Workspace W, SweepArgs a)
{
    static_assert(NQ % NT == 0 && (NT == 1 || NT == 2 || NT == 3), "threads per band");
    constexpr int G = NQ / NT, NG = 4 * G, NC = RRLW_SWEEPC_CODES;
    constexpr bool DOWN = PHASE != 2, UP = PHASE != 1, TWO = PHASE == 2;       // TWO: total and clear-sky streams differ
    constexpr int NVAL = sweepc_nval(PHASE, IDRV);
    extern __shared__ __align__(16) unsigned char smem[];
    // (a wave = 64 consecutive threadIdx.x of one (y, z): band and sub-block are wave-uniform - made scalar, so that everything derived
    // from the band, buffer descriptors included, lives in scalar registers)
    const int tx = threadIdx.x, ty = __builtin_amdgcn_readfirstlane(threadIdx.y), sub = __builtin_amdgcn_readfirstlane(threadIdx.z);
    const int bi = ty / NT, part = ty % NT;                                     // band of the group, part of the band
    const int ny = blockDim.y, nb = ny / NT, nsb = blockDim.z, ncw = 64 * nsb;  // waves per column sub-block, bands of the group, sub-blocks, columns of the workgroup
    const float2 *s_lut = reinterpret_cast<const float2 *>(smem);
    double (*s_pl)[184] = reinterpret_cast<double (*)[184]>(smem + SWEEP_LUT_BYTES + bi * SWEEPC_BAND_BYTES);
    double (*s_fr)[16] = reinterpret_cast<double (*)[16]>(smem + SWEEP_LUT_BYTES + bi * SWEEPC_BAND_BYTES + SWEEP_PL_BYTES);
    double *red = reinterpret_cast<double *>(smem + SWEEP_LUT_BYTES + nb * SWEEPC_BAND_BYTES);      // [2][NVAL][nb][NC][ncw]
    // the wave's 64-column block: position (workgroup, sub-block) of the order k_blocksort left (deepest clouds first); a cloud-free call
    // (PHASE 0) has no order and takes the blocks as they come
    const int slot = blockIdx.x * nsb + sub;
    const int cblock = PHASE == 0 ? slot : __builtin_amdgcn_readfirstlane(W.order[slot]);
    const int col = cblock * 64 + tx;
    const int bsel = a.split ? (int)blockIdx.y : 0;                             // split: this workgroup's band of the group
    const unsigned long long bands = a.bands >> (4 * bsel);
    const int B = (int)((bands >> (4 * bi)) & 15ull) + 1;
    const bool incol = col < a.ncol;
    const int colc = incol ? col : a.ncol - 1;
    const int quad = __builtin_amdgcn_readfirstlane(band_qstart(B) + part * G);
    const int g0 = NG * part;                                                   // first of this thread's g-points within the band
    const int pc = pcol(W, colc);                                               // the column at this position (caller arrays)
    const size_t gc = (size_t)a.col0 + pc;
    const int nlay = W.nlay, nct = a.nct;
    const size_t ncb = W.ncolb;
    const bool alt16 = (B == 16 && a.istart == 16);
    const bool lo_bin = (LO_BINARY >> (B - 1)) & 1u, up_bin = (UP_BINARY >> (B - 1)) & 1u;
    const bool any_bin = lo_bin || up_bin;
    const int base_up = ((UP_ZERO >> (B - 1)) & 1u) ? 14 : (((UP_FROM_A >> (B - 1)) & 1u) ? 0 : 9);
    sweep_stage_lut(T, smem, (sub * ny + ty) * 64 + tx, 64 * ny * nsb);
    sweep_stage_band(T, s_pl, s_fr, B, alt16, lo_bin, up_bin, (sub * NT + part) * 64 + tx, NT * ncw);
    __syncthreads();
    const int lo = PHASE == 0 ? 1 : __builtin_amdgcn_readfirstlane(W.hgrp[(blockIdx.x * nsb) / SORT_GROUP]) + 1;       // layers lo .. nlay (uniform over the workgroup)
    const size_t qstride = (size_t)nlay * ncb;
    const unsigned *__restrict__ sC = W.scr[S_CODE] + (size_t)quad * qstride * CODE_WORDS;
    const unsigned *__restrict__ sFw = W.fw + (size_t)fw_slot(B) * nlay * ncb;
    // (temperatures: the caller's rows - or, for a window whose columns are taken in another order, their copies in position order: the
    // lane's offset is its position either way, eight consecutive bytes per lane)
    const bool moved = pmoved(W, colc);
    const double *__restrict__ tlay = moved ? W.tlayc : a.tlay + a.col0;
    const double *__restrict__ tlev = moved ? W.tlevc : a.tlev + a.col0;
    const size_t tstride = moved ? ncb : (size_t)nct;
    const unsigned offc = (unsigned)colc * (unsigned)CODE_BYTES, off8p = (unsigned)colc * 8u, off4 = (unsigned)colc * 4u;
    const size_t pcb = (size_t)W.pcb;
    const size_t gslab = (size_t)(a.group + bsel) * (nlay + 1) * pcb;        // (uniform: the group's slabs; a lane's column is the 32-bit offset of a buffer store)
    double *__restrict__ gdn1 = W.gdn1 + gslab;
    double *__restrict__ gup1 = W.gup1 + gslab;
    Part2 *__restrict__ gup = W.gup + gslab;
    Part2 *__restrict__ gdp = W.gdp + gslab;
    const unsigned so8 = (unsigned)col * 8u, so16 = (unsigned)col * 16u;
    const int laytrop = W.laytrop[colc];
    const double *tp0 = s_pl[0], *tp1 = s_pl[1];
    double2 *hand = reinterpret_cast<double2 *>(W.hand) + ((size_t)quad * ncb + colc) * 2;      // [stream][quad][column][2]
    const size_t hstream = (size_t)NQUAD * ncb * 2;

    // loads of layer `lev` (any integer: rows outside 1 .. nlay are clamped, their values never used); zoff selects the interface whose
    // temperature the sweep direction needs: 0 = level lev - 1 (below the layer, downward sweep), 1 = level lev (above, upward sweep)
    // (rows as scalar offsets - bload_*'s soff - in the two-stream phases, which are bound by instruction issue: -1.5 %.  The downward
    // phase above the clouds streams at the memory pipeline's rate and lives by WHERE its code loads sit in the level body; with the
    // lighter address arithmetic the scheduler gathers them at the end of the body, the prefetch distance shrinks and the phase takes 3 %
    // longer (profiles/round5_instruction_diet.md): it keeps the row addresses it had.)
    constexpr bool SOFF = PHASE != 1;
    const unsigned ncb32 = (unsigned)ncb, tstr32 = (unsigned)tstride;
    auto fill_t = [&](auto bin_tag, int lev, int zoff, SweepcLev<G> &q) {
        const int l = min(max(lev, 1), nlay);
        if constexpr (SOFF) {
            const unsigned r = (unsigned)(l - 1);
            q.tl = bload_f64(row_ptr(tlay, r, tstr32), off8p);
            q.tz = bload_f64(row_ptr(tlev, r + (unsigned)zoff, tstr32), off8p);
            if constexpr (decltype(bin_tag)::value) q.w = bload_u32(sFw, off4, r * ncb32 * 4u);
        } else {
            q.tl = bload_f64(tlay + tstride * (l - 1), off8p);
            q.tz = bload_f64(tlev + tstride * (l - 1 + zoff), off8p);
            if constexpr (decltype(bin_tag)::value) q.w = bload_u32(sFw + (size_t)(l - 1) * ncb, off4);
        }
    };
    auto ld_c = [&](int lev, int k) -> pk4 {
        const int l = min(max(lev, 1), nlay);
        if constexpr (SOFF) return bload_pk4_nt(sC + (size_t)k * qstride * CODE_WORDS, offc, (unsigned)(l - 1) * ncb32 * (unsigned)CODE_BYTES);
        else return bload_pk4_nt(sC + ((size_t)k * qstride + (size_t)(l - 1) * ncb) * CODE_WORDS, offc);
    };
    // fraction rows of layer `lev`: first row and interpolation weight (taumol :556-561, :692-693)
    auto frac_row = [&](int lev, unsigned fwv, double &fpl) -> const double * {
        const bool lower = lev <= laytrop;
        if (any_bin) {          // uniform
            const unsigned w = (lower ? lo_bin : up_bin) ? fwv : 0x10000000u;
            const int r0 = clampi((lower ? 0 : base_up) + (int)(w >> 28) - 1, 0, 14);
            fpl = (double)(w & 0x0fffffffu) * (1.0 / 268435456.0);
            return &s_fr[r0][g0];
        }
        fpl = 0.0;
        return &s_fr[lower ? 0 : base_up][g0];
    };

    // A wave's raw sum over its g-points of a level goes to the reduction buffer [buffer][value][wave][slot][column]; after the NC levels of a
    // round the workgroup meets at ONE barrier, wave `ty` forms the bands' partials of slot ty (ty + ny, ..) - (sum of the band's waves x 0.5)
    // x delwave, rtrn :549-562 - adds them in band-list order and stores the group's partial.  Two buffers: the next round writes the other
    // one, and a buffer is written again only after the barrier in between, which no wave passes before every wave has finished reading it.
    const unsigned rlane = (unsigned)(sub * 64 + tx);
    const unsigned vstride = (unsigned)(ny * NC * ncw), rband = (unsigned)(ty * NC * ncw);      // (uniform)
    unsigned bufoff = 0u;                                                                       // 0 / NVAL * vstride
    auto red_put = [&](int c, int val, double v) { red[bufoff + val * vstride + rband + (unsigned)(c * ncw) + rlane] = v; };
    auto round_end = [&](auto dn_tag, int lev0, int nvalid) {
#pragma clang fp contract(off)      // (a band's weight, the flux factor and the sum over the bands round as written: in k_sweepz too - a level's partial must not depend on the kernel that swept it)
        constexpr bool DN = decltype(dn_tag)::value;
        constexpr int NV = DN ? 1 : NVAL;
        __syncthreads();
        for (int c = ty; c < nvalid; c += ny) {       // (wave-uniform)
            const int lvl = DN ? lev0 - c - 1 : lev0 + c;
            double sv[NV];
#pragma unroll
            for (int val = 0; val < NV; val++) {
                const double *r = red + (bufoff + val * vstride + (unsigned)(c * ncw) + rlane);
                const bool deriv = IDRV && !DN && val >= (TWO ? 2 : 1);
                double sum = 0.0;
                for (int q = 0; q < nb; q++) {
                    double pq = r[(unsigned)(q * NT * NC * ncw)];            // (the band's parts in the order one thread adds its quads: pairs first)
                    if constexpr (NT >= 2) pq = pq + r[(unsigned)((q * NT + 1) * NC * ncw)];
                    if constexpr (NT == 3) pq = pq + r[(unsigned)((q * NT + 2) * NC * ncw)];
                    double v = (pq * 0.5) * T.delwave[(int)((bands >> (4 * q)) & 15ull)];
                    if (deriv) v = v * T.fluxfac;
                    sum = q == 0 ? v : sum + v;
                }
                sv[val] = sum;
            }
            if (incol) {
                if constexpr (DN) bstore_f64(gdn1 + (size_t)lvl * pcb, so8, sv[0]);
                else {
                    if constexpr (TWO) bstore_f64x2(gup + (size_t)lvl * pcb, so16, sv[0], sv[1]);
                    else bstore_f64(gup1 + (size_t)lvl * pcb, so8, sv[0]);
                    if constexpr (IDRV) bstore_f64x2(gdp + (size_t)lvl * pcb, so16, TWO ? sv[2] : sv[1], TWO ? sv[3] : sv[1]);
                }
            }
        }
        bufoff = bufoff ? 0u : NVAL * vstride;
    };

    double rad[NG], radc[TWO ? NG : 1], drad[IDRV ? NG : 1], dradc[(IDRV && TWO) ? NG : 1];
#pragma unroll
    for (int j = 0; j < NG; j++) rad[j] = 0.0;

    // One level: layer `lev`, its temperatures in `cur`, its codes in cc[.][lev % NC]; dir = -1 downward (Planck difference towards the
    // interface below, partial of level lev - 1), +1 upward.  BIN: the band's Planck fractions are interpolated between two rows.
    // The slot pieces are re-issued for level lev + NC * dir (codes) / lev + dir (temperatures) as soon as they have been consumed.
    SweepcLev<G> cur;
    cur.w = 0u;
    pk4 cc[G][NC];
    int ki = 0;
    auto level = [&](auto bin_tag, auto dn_tag, int lev, int slot) {
        constexpr bool BIN = decltype(bin_tag)::value, DN = decltype(dn_tag)::value;
        constexpr int dir = DN ? -1 : 1;
        ki_salu(ki);
        double fpl;
        const double *row = frac_row(lev, cur.w, fpl);
        const double blay = planck_at(tp0, tp0, cur.tl);
        const double dpl = planck_at(tp0, (DN && alt16 && lev == 1) ? tp1 : tp0, cur.tz) - blay;
        fill_t(bin_tag, lev + dir, DN ? 0 : 1, cur);
        // (sums as a tree - pairs inside a quad, then the quads: a running sum over 16 g-points is a chain of 16 dependent additions, and a
        // wave issues in order, so each of them would hold the wave for the latency of the one before)
        double qs[G], qsc[G], qd[G], qdc[G];
#pragma unroll
        for (int k = 0; k < G; k++) {
            float2 e[4];
            pk4 pk;
#pragma unroll
            for (int c = 0; c < NC; c++) if (c == slot) pk = cc[k][c];
            const scr4 ck = unpack4(pk);
#pragma unroll
            for (int jj = 0; jj < 4; jj++) e[jj] = RRLW_LUT_ENTRY(s_lut, code_index(ck.v[jj]));
            {
                const pk4 nx = ld_c(lev + NC * dir, k);
#pragma unroll
                for (int c = 0; c < NC; c++) if (c == slot) cc[k][c] = nx;
            }
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = 4 * k + jj;
                double atr, tfn;
                decode(ck.v[jj], e[jj], atr, tfn);
                double fr = row[j];
                if constexpr (BIN) fr = fma(fpl, row[16 + j] - fr, fr);
                const double bb = fr * fma(tfn, dpl, blay);
                rad[j] = fma(bb - rad[j], atr, rad[j]);
                if constexpr (TWO && !DN) radc[j] = fma(bb - radc[j], atr, radc[j]);
                if constexpr (IDRV && !DN) {
                    const double tr = one_minus(atr);
                    drad[j] = times(drad[j], tr);
                    if constexpr (TWO) dradc[j] = times(dradc[j], tr);
                }
            }
            qs[k] = (rad[4 * k] + rad[4 * k + 1]) + (rad[4 * k + 2] + rad[4 * k + 3]);
            if constexpr (TWO && !DN) qsc[k] = (radc[4 * k] + radc[4 * k + 1]) + (radc[4 * k + 2] + radc[4 * k + 3]);
            if constexpr (IDRV && !DN) {
                qd[k] = (drad[4 * k] + drad[4 * k + 1]) + (drad[4 * k + 2] + drad[4 * k + 3]);
                if constexpr (TWO) qdc[k] = (dradc[4 * k] + dradc[4 * k + 1]) + (dradc[4 * k + 2] + dradc[4 * k + 3]);
            }
            if (RRLW_SWEEPC_QUAD_BARRIER) __builtin_amdgcn_sched_barrier(0);
        }
        auto tree = [&](const double (&q)[G]) -> double {
            if constexpr (G == 1) return q[0];
            else if constexpr (G == 2) return q[0] + q[1];
            else if constexpr (G == 3) return (q[0] + q[1]) + q[2];
            else return (q[0] + q[1]) + (q[2] + q[3]);
        };
        red_put(slot, 0, tree(qs));
        if constexpr (TWO && !DN) red_put(slot, 1, tree(qsc));
        if constexpr (IDRV && !DN) {
            red_put(slot, TWO ? 2 : 1, tree(qd));
            if constexpr (TWO) red_put(slot, 3, tree(qdc));
        }
    };
    auto sweep = [&](auto bin_tag, auto dn_tag) {
        constexpr bool DN = decltype(dn_tag)::value;
        constexpr int dir = DN ? -1 : 1;
        const int first = DN ? nlay : lo, count = nlay - lo + 1;
        // (issued in the order the steady state has them in flight at the top of a level - codes of the slots that are older than the
        // level's temperatures, the temperatures, the youngest slot - so that the waits of the loop header can be counted ones)
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (c == NC - 1) fill_t(bin_tag, first, DN ? 0 : 1, cur);
#pragma unroll
            for (int k = 0; k < G; k++) cc[k][c] = ld_c(first + c * dir, k);
        }
        int n = 0;
        for (; n + NC <= count; n += NC) {          // whole rounds of NC levels: a straight-line body (a skipped level inside the loop makes
#pragma unroll                                      // the compiler rotate the slot registers through copies, which wait for every load)
            for (int c = 0; c < NC; c++) level(bin_tag, dn_tag, first + (n + c) * dir, c);
            round_end(dn_tag, first + n * dir, NC);
        }
        const int rem = count - n;                  // (uniform over the workgroup)
#pragma unroll
        for (int c = 0; c < NC - 1; c++) {
            if (c < rem) level(bin_tag, dn_tag, first + (n + c) * dir, c);
        }
        if (rem > 0) round_end(dn_tag, first + n * dir, rem);
    };
    using std::true_type;
    using std::false_type;

    if constexpr (DOWN) {
        // ------------------------------------------------------------------ downward: layers nlay .. lo
        if (incol && ty == 0) bstore_f64(gdn1 + (size_t)nlay * pcb, so8, 0.0);
        if (any_bin) sweep(true_type{}, true_type{}); else sweep(false_type{}, true_type{});
    }
    if constexpr (PHASE == 1) {                     // downward radiances at level ltop for k_sweepz
        if (incol) {
#pragma unroll
            for (int k = 0; k < G; k++) {
                hand[(size_t)k * ncb * 2] = make_double2(rad[4 * k], rad[4 * k + 1]);
                hand[(size_t)k * ncb * 2 + 1] = make_double2(rad[4 * k + 2], rad[4 * k + 3]);
            }
        }
        return;
    }

    if constexpr (PHASE == 0) {
        // ------------------------------------------------------------------ surface: rtrn :476-495
        const double reflect = 1. - a.emis[gc + (size_t)nct * (B - 1)];
        const double pb = W.percol[(size_t)(PC_PLANKBND + B - 1) * ncb + colc];
        const double dpb = IDRV ? W.percol[(size_t)(PC_DPLANKBND + B - 1) * ncb + colc] : 0.0;
        double fpl;
        const double *row = frac_row(1, any_bin ? bload_u32(sFw, off4) : 0u, fpl);        // the surface emits with the lowest layer's Planck fractions
        double usum = 0.0, dusum = 0.0;
#pragma unroll
        for (int j = 0; j < NG; j++) {
            const double fr = any_bin ? fma(fpl, row[16 + j] - row[j], row[j]) : row[j];
            rad[j] = fma(reflect, rad[j], fr * pb);
            usum = usum + rad[j];
            if constexpr (IDRV) { drad[j] = times(fr, dpb); dusum = dusum + drad[j]; }
        }
        red_put(0, 0, usum);
        if constexpr (IDRV) red_put(0, 1, dusum);
        round_end(false_type{}, 0, 1);              // level 0
    } else {                                        // upward radiances at level ltop from k_sweepz
#pragma unroll
        for (int k = 0; k < G; k++) {
            const double2 *h = hand + (size_t)k * ncb * 2;
            const double2 a0 = h[hstream], a1 = h[hstream + 1], c0 = h[2 * hstream], c1 = h[2 * hstream + 1];
            rad[4 * k] = a0.x; rad[4 * k + 1] = a0.y; rad[4 * k + 2] = a1.x; rad[4 * k + 3] = a1.y;
            if constexpr (TWO) { radc[4 * k] = c0.x; radc[4 * k + 1] = c0.y; radc[4 * k + 2] = c1.x; radc[4 * k + 3] = c1.y; }
            if constexpr (IDRV) {
                const double2 d0 = h[3 * hstream], d1 = h[3 * hstream + 1], e0 = h[4 * hstream], e1 = h[4 * hstream + 1];
                drad[4 * k] = d0.x; drad[4 * k + 1] = d0.y; drad[4 * k + 2] = d1.x; drad[4 * k + 3] = d1.y;
                if constexpr (TWO) { dradc[4 * k] = e0.x; dradc[4 * k + 1] = e0.y; dradc[4 * k + 2] = e1.x; dradc[4 * k + 3] = e1.y; }
            }
        }
    }
    if constexpr (UP) {
        // ------------------------------------------------------------------ upward: layers lo .. nlay
        if (any_bin) sweep(true_type{}, false_type{}); else sweep(false_type{}, false_type{});
    }
}

// ------------------------------------------------------------------------------------------------
// k_sweepz : the cloud zone (layers 1 .. ltop) of the maximum-random-overlap sweep, rtrnmr (src/rrtmg_lw_rtrnmr.f90:509-704), without
//            d(flux)/dT, in k_sweepc's form: one thread = (column, G quads of a band), a workgroup = the bands of one group, every load
//            issued on every path (the cloudy level's extra inputs - total-optical-depth codes, cloud fraction, overlap factors - are
//            read for every level of the zone) so that the waits are counted ones, partials added over the group in LDS every two
//            levels, ONE barrier per round.  A wave none of whose 64 columns is cloudy at a level runs the clear-sky body; otherwise
//            both updates are formed and selected per lane (what divergence does anyway).  Its predecessor k_sweep (one workgroup per band, removed) did the same work at
//            1.9x the time per clear level (per-quad duplication of the level's Planck terms, conditional loads with full waits, two
//            barriers per four levels).
// ------------------------------------------------------------------------------------------------
#ifndef RRLW_SWEEPZ_CT_SLOTS
#define RRLW_SWEEPZ_CT_SLOTS 1
#endif
#ifndef RRLW_SWEEPZ_G2
#define RRLW_SWEEPZ_G2 0          // 1: two quads per thread for bands of 4 and 2 quads (253 registers, two waves per SIMD: measured slower)
#endif
__host__ __device__ constexpr int sweepz_g(int NQ) { return (RRLW_SWEEPZ_G2 && (NQ == 4 || NQ == 2)) ? 2 : 1; }          // quads per thread
__host__ __device__ constexpr int sweepz_nt(int NQ) { return NQ / sweepz_g(NQ); }                    // threads (waves) per band
#ifndef RRLW_SWEEPZ_WAVES_G2
#define RRLW_SWEEPZ_WAVES_G2 2
#endif
#ifndef RRLW_SWEEPZ_WAVES_G1
#define RRLW_SWEEPZ_WAVES_G1 3
#endif
#ifndef RRLW_SWEEPZ_WAVES_IDRV
#define RRLW_SWEEPZ_WAVES_IDRV 2          // with d(flux)/dT: 16 more state registers per thread
#endif
__host__ __device__ constexpr int sweepz_waves(int NQ, bool IDRV = false)
{
    return IDRV ? RRLW_SWEEPZ_WAVES_IDRV : (sweepz_g(NQ) == 2 ? RRLW_SWEEPZ_WAVES_G2 : RRLW_SWEEPZ_WAVES_G1);
}
__host__ __device__ constexpr int sweepz_lds_bytes(int nb, int nsb, int NT, bool IDRV = false)
{
    return SWEEP_LUT_BYTES + nb * SWEEPC_BAND_BYTES + 2 * (IDRV ? 4 : 2) * nb * NT * RRLW_SWEEPC_CODES * nsb * 64 * 8;
}
__host__ __device__ constexpr int sweepz_nsb(int NQ, int nb, bool IDRV = false)
{
    const int nt = sweepz_nt(NQ);
    int nsb = nsb_fit(4 * sweepz_waves(NQ, IDRV) / (nb * nt));
    while (nsb > 1 && sweepz_lds_bytes(nb, nsb, nt, IDRV) > SWEEPC_LDS_MAX) nsb = nsb_fit(nsb - 1);
    return nsb;
}

// bands per group: wave slots and LDS of one workgroup
__host__ __device__ constexpr int sweepz_group_cap(int NQ, bool IDRV)
{
    const int nt = sweepz_nt(NQ);
    int cap = 4 * sweepz_waves(NQ, IDRV) / nt;
    if (cap < 1) cap = 1;
    while (cap > 1 && sweepz_lds_bytes(cap, 1, nt, IDRV) > SWEEPC_LDS_MAX) cap--;
    return cap;
}

struct SweepzLev { double tl, tz, cf; unsigned w, flag; };

// MODE: 1 rtrn (random overlap), 2 rtrnmr, 3 rtrnmc with per-g-point arrays, 4 rtrnmc with the generator's sub-column mask
// IDRV: d(upward flux)/dT carried along (idrv = 1), src/rrtmg_lw_rtrnmr.f90:655-703 and its siblings
template <int NQ, int MODE, bool IDRV = false>
__global__ __launch_bounds__(256 * sweepz_waves(NQ, IDRV), sweepz_waves(NQ, IDRV)) void k_sweepz(DevTables T, Workspace W, SweepArgs a)
{
    static_assert(MODE >= 1 && MODE <= 4, "modes of the cloud-zone sweep");
    constexpr int G = sweepz_g(NQ), NT = sweepz_nt(NQ), NG = 4 * G, NC = RRLW_SWEEPC_CODES, NVAL = IDRV ? 4 : 2;
    constexpr int NS2 = MODE == 2 ? NG : 1;          // rtrnmr's extra state
    constexpr int NSD = IDRV ? NG : 1;               // d/dT state
    extern __shared__ __align__(16) unsigned char smem[];
    const int tx = threadIdx.x, ty = __builtin_amdgcn_readfirstlane(threadIdx.y), sub = __builtin_amdgcn_readfirstlane(threadIdx.z);
    const int bi = ty / NT, part = ty % NT;
    const int ny = blockDim.y, nb = ny / NT, nsb = blockDim.z, ncw = 64 * nsb;
    const float2 *s_lut = reinterpret_cast<const float2 *>(smem);
    double (*s_pl)[184] = reinterpret_cast<double (*)[184]>(smem + SWEEP_LUT_BYTES + bi * SWEEPC_BAND_BYTES);
    double (*s_fr)[16] = reinterpret_cast<double (*)[16]>(smem + SWEEP_LUT_BYTES + bi * SWEEPC_BAND_BYTES + SWEEP_PL_BYTES);
    double *red = reinterpret_cast<double *>(smem + SWEEP_LUT_BYTES + nb * SWEEPC_BAND_BYTES);      // [2][NVAL][ny][NC][ncw]
    const int cblock = __builtin_amdgcn_readfirstlane(W.order[blockIdx.x * nsb + sub]);       // the wave's 64-column block (see k_sweepc)
    const int col = cblock * 64 + tx;
    const int bsel = a.split ? (int)blockIdx.y : 0;                             // (see k_sweepc)
    const unsigned long long bands = a.bands >> (4 * bsel);
    const int B = (int)((bands >> (4 * bi)) & 15ull) + 1;
    const bool incol = col < a.ncol;
    const int colc = incol ? col : a.ncol - 1;
    const int quad = __builtin_amdgcn_readfirstlane(band_qstart(B) + part * G);
    const int g0 = NG * part;
    const int pc = pcol(W, colc);                                               // (see k_sweepc)
    const size_t gc = (size_t)a.col0 + pc;
    const int nlay = W.nlay, nct = a.nct;
    const size_t ncb = W.ncolb;
    const bool alt16 = (B == 16 && a.istart == 16);
    const bool lo_bin = (LO_BINARY >> (B - 1)) & 1u, up_bin = (UP_BINARY >> (B - 1)) & 1u;
    const bool any_bin = lo_bin || up_bin;
    const int base_up = ((UP_ZERO >> (B - 1)) & 1u) ? 14 : (((UP_FROM_A >> (B - 1)) & 1u) ? 0 : 9);
    sweep_stage_lut(T, smem, (sub * ny + ty) * 64 + tx, 64 * ny * nsb);
    sweep_stage_band(T, s_pl, s_fr, B, alt16, lo_bin, up_bin, (sub * NT + part) * 64 + tx, NT * ncw);
    __syncthreads();
    const int ltop = __builtin_amdgcn_readfirstlane(W.hgrp[(blockIdx.x * nsb) / SORT_GROUP]);       // layers 1 .. ltop (uniform over the workgroup)
    const size_t qstride = (size_t)nlay * ncb;
    const unsigned *__restrict__ sC = W.scr[S_CODE] + (size_t)quad * qstride * CODE_WORDS;
    const unsigned *__restrict__ sCt = W.scr[S_CODET] + (size_t)quad * qstride * CODE_WORDS;
    const unsigned *__restrict__ sFw = W.fw + (size_t)fw_slot(B) * nlay * ncb;
    const int *__restrict__ sFlag = W.cflag;
    const bool moved = pmoved(W, colc);                                         // (see k_sweepc)
    const double *__restrict__ tlay = moved ? W.tlayc : a.tlay + a.col0;
    const double *__restrict__ tlev = moved ? W.tlevc : a.tlev + a.col0;
    const double *__restrict__ cldf = moved ? W.cldfc : a.cldfrac + a.col0;
    const size_t tstride = moved ? ncb : (size_t)nct;
    const unsigned off16 = (unsigned)colc * 16u, offc = (unsigned)colc * (unsigned)CODE_BYTES, off8 = (unsigned)colc * 8u, off4 = (unsigned)colc * 4u;
    const size_t pcb = (size_t)W.pcb;
    const size_t gslab = (size_t)(a.group + bsel) * (nlay + 1) * pcb;
    Part2 *__restrict__ gdn = W.gdn + gslab;
    Part2 *__restrict__ gup = W.gup + gslab;
    Part2 *__restrict__ gdp = W.gdp + gslab;
    const unsigned so16 = (unsigned)col * 16u;
    const int laytrop = W.laytrop[colc];
    const double *tp0 = s_pl[0], *tp1 = s_pl[1];
    double2 *hand = reinterpret_cast<double2 *>(W.hand) + ((size_t)quad * ncb + colc) * 2;
    const size_t hstream = (size_t)NQUAD * ncb * 2;
    const bool colcloud = (bload_u32(sFlag, off4) & 8u) != 0;
    // MODE 4: the mask bits of this thread's g-points (bits of padding g-points cleared): first g-point, its words, valid count
    const int ig0 = band_g0(B) + g0, mw0 = ig0 >> 5, mw1 = mw0 < 4 ? mw0 + 1 : 4;
    const int nvalid = max(0, min(NG, band_ng(B) - g0));

    auto clampl = [&](int lev) { return min(max(lev, 1), nlay); };
    // the level's own inputs, one level ahead; zoff: 0 = interface below the layer (downward), 1 = above (upward)
    // (LITE: a level below the group's lowest cloud - no column of the workgroup is cloudy there: its cloud fraction and flag word are not read)
    const unsigned ncb32 = (unsigned)ncb, tstr32 = (unsigned)tstride;
    auto fill_t = [&](auto bin_tag, auto lite_tag, int lev, int zoff, SweepzLev &q) {
        const unsigned r = (unsigned)(clampl(lev) - 1);
        q.tl = bload_f64(row_ptr(tlay, r, tstr32), off8);
        q.tz = bload_f64(row_ptr(tlev, r + (unsigned)zoff, tstr32), off8);
        if constexpr (!decltype(lite_tag)::value) {
            q.cf = bload_f64(row_ptr(cldf, r, tstr32), off8);
            q.flag = bload_u32(sFlag, off4, (r + 1u) * ncb32 * 4u);
        }
        if constexpr (decltype(bin_tag)::value) q.w = bload_u32(sFw, off4, r * ncb32 * 4u);
    };
    auto ld_c = [&](const unsigned *base, int lev, int k) -> pk4 {
        return bload_pk4_nt(base + (size_t)k * qstride * CODE_WORDS, offc, (unsigned)(clampl(lev) - 1) * ncb32 * (unsigned)CODE_BYTES);
    };
    auto frac_row = [&](int lev, unsigned fwv, double &fpl) -> const double * {
        const bool lower = lev <= laytrop;
        if (any_bin) {          // uniform
            const unsigned w = (lower ? lo_bin : up_bin) ? fwv : 0x10000000u;
            const int r0 = clampi((lower ? 0 : base_up) + (int)(w >> 28) - 1, 0, 14);
            fpl = (double)(w & 0x0fffffffu) * (1.0 / 268435456.0);
            return &s_fr[r0][g0];
        }
        fpl = 0.0;
        return &s_fr[lower ? 0 : base_up][g0];
    };

    // reduction over the group, as in k_sweepc (raw sums in, band weight applied by the reducing wave)
    const unsigned rlane = (unsigned)(sub * 64 + tx);
    const unsigned vstride = (unsigned)(ny * NC * ncw), rband = (unsigned)(ty * NC * ncw);
    unsigned bufoff = 0u;
    auto red_put = [&](int c, int val, double v) { red[bufoff + val * vstride + rband + (unsigned)(c * ncw) + rlane] = v; };
    auto round_end = [&](auto dn_tag, int lev0, int nvalid) __attribute__((always_inline)) {
#pragma clang fp contract(off)      // (as in k_sweepc)
        constexpr bool DN = decltype(dn_tag)::value;
        __syncthreads();
        constexpr int NV = DN ? 2 : NVAL;
        for (int c = ty; c < nvalid; c += ny) {       // (wave-uniform)
            const int lvl = DN ? lev0 - c - 1 : lev0 + c;
            double sv[NV];
#pragma unroll
            for (int val = 0; val < NV; val++) {
                const double *r = red + (bufoff + val * vstride + (unsigned)(c * ncw) + rlane);
                double sum = 0.0;
                for (int q = 0; q < nb; q++) {
                    // (the band's quads in k_sweepc's order - pairs first: a level's partial must not depend on which of the two kernels swept it,
                    // i.e. on the hand-off level its block happened to get)
                    double pq = r[(unsigned)(q * NT * NC * ncw)];
                    if constexpr (NT >= 2) pq = pq + r[(unsigned)((q * NT + 1) * NC * ncw)];
                    if constexpr (NT == 3) pq = pq + r[(unsigned)((q * NT + 2) * NC * ncw)];
                    if constexpr (NT == 4) pq = pq + (r[(unsigned)((q * NT + 2) * NC * ncw)] + r[(unsigned)((q * NT + 3) * NC * ncw)]);
                    double v = (pq * 0.5) * T.delwave[(int)((bands >> (4 * q)) & 15ull)];
                    if (val >= 2) v = v * T.fluxfac;
                    sum = q == 0 ? v : sum + v;
                }
                sv[val] = sum;
            }
            if (incol) {
                bstore_f64x2((DN ? gdn : gup) + (size_t)lvl * pcb, so16, sv[0], sv[1]);
                if constexpr (IDRV && !DN) bstore_f64x2(gdp + (size_t)lvl * pcb, so16, sv[2], sv[3]);
            }
        }
        bufoff = bufoff ? 0u : NVAL * vstride;
    };

    // state: total-sky radiance, clear-sky radiance, rtrnmr's cloudy / clear parts and carried correction, per g-point
    double rad[NG], radc[NG], cldrad[NS2], clrrad[NS2], radmr[NS2], drad[NSD], dradc[NSD];
#pragma unroll
    for (int j = 0; j < NS2; j++) { cldrad[j] = 0.0; clrrad[j] = 0.0; radmr[j] = 0.0; }
    {                               // downward radiances at level ltop from k_sweepc<., 1>; clear = total up there
#pragma unroll
        for (int k = 0; k < G; k++) {
            const double2 h0 = hand[(size_t)k * ncb * 2], h1 = hand[(size_t)k * ncb * 2 + 1];
            rad[4 * k] = h0.x; rad[4 * k + 1] = h0.y; rad[4 * k + 2] = h1.x; rad[4 * k + 3] = h1.y;
        }
        if (ltop == nlay) {         // the zone reaches the top of the column: nothing comes down into it (rtrn :352 radld = 0); with the zone
#pragma unroll                      // kernel walking all levels (k_blocksort, force_top) no clear-sky launch has written the hand-off array
            for (int j = 0; j < NG; j++) rad[j] = 0.0;
            if (incol && ty == 0) bstore_f64(W.gdn1 + gslab + (size_t)nlay * pcb, (unsigned)col * 8u, 0.0);      // downward flux at the top level
        }
#pragma unroll
        for (int j = 0; j < NG; j++) radc[j] = rad[j];
    }
    bool seen = false;              // downward: a cloud lies above (iclddn); upward: the column holds cloud (set before the upward sweep)

    SweepzLev cur;
    cur.w = 0u;
    // code slots: NC levels of the gas codes in flight, NCT of the total (gas + cloud) ones (a second slot of those put the rtrnmr
    // instantiations 3-5 registers over their 168: scratch reloads count in vmcnt and drain the prefetches)
    constexpr int NCT = RRLW_SWEEPZ_CT_SLOTS;
    static_assert(NCT == 1 || NCT == NC, "slots of the total-optical-depth codes");
    pk4 cc[G][NC], ct[G][NCT];
    // One level.  DN: downward (Planck difference towards the interface below, partial of level lev - 1, istcldd = flag bit 1), else
    // upward (istcld = bit 2).  The overlap factors of the level are requested first and used last.
    int ki = 0;
    auto level = [&](auto bin_tag, auto dn_tag, int lev, int slot) __attribute__((always_inline)) {
        constexpr bool BIN = decltype(bin_tag)::value, DN = decltype(dn_tag)::value;
        constexpr int dir = DN ? -1 : 1;
        ki_salu(ki);
        double2 f0 = make_double2(0., 0.), f1 = f0, f2 = f0;
        double efcl = 0.0;
        unsigned mlo = 0u, mhi = 0u;
        float4 cf4[G], ef4[G];
        if constexpr (MODE == 2) {
            const double2 *ov = W.ovl + (size_t)(DN ? 0 : nlay + 1) * 3 * ncb;            // (the direction's half: uniform, fixed)
            const unsigned so = (unsigned)clampl(lev) * 3u * ncb32 * 16u;
            f0 = bload_f64x2(ov, off16, so); f1 = bload_f64x2(ov, off16, so + ncb32 * 16u); f2 = bload_f64x2(ov, off16, so + ncb32 * 32u);
        } else {
            if constexpr (MODE != 3) efcl = bload_f64(W.efcl + (size_t)(B - 1) * nlay * ncb, off8, (unsigned)(clampl(lev) - 1) * ncb32 * 8u);
            if constexpr (MODE == 3) {              // cloud fraction (0 / 1) and effective emissivity per g-point: 8 floats per quad
#pragma unroll
                for (int k = 0; k < G; k++) {
                    const float *pc = W.cfef + (size_t)(quad + k) * nlay * ncb * 8;
                    const unsigned so = (unsigned)(clampl(lev) - 1) * ncb32 * 32u;
                    const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(sweep_rsrc(pc), (int)((unsigned)colc * 32u), (int)so, 0);
                    const u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(sweep_rsrc(pc), (int)((unsigned)colc * 32u + 16u), (int)so, 0);
                    __builtin_memcpy(&cf4[k], &c, 16);
                    __builtin_memcpy(&ef4[k], &e, 16);
                }
            }
            if constexpr (MODE == 4) {              // the two mask words that can hold this thread's g-points
                const unsigned moff = (unsigned)(W.mask_col0 + gc) * 4u;
                const unsigned *mrow = W.mask + (size_t)(clampl(lev) - 1) * W.mask_stride;
                mlo = bload_u32(mrow + (size_t)mw0 * nlay * W.mask_stride, moff);
                mhi = bload_u32(mrow + (size_t)mw1 * nlay * W.mask_stride, moff);
            }
        }
        double fpl;
        const double *row = frac_row(lev, cur.w, fpl);
        const double blay = planck_at(tp0, tp0, cur.tl);
        const double dpl = planck_at(tp0, (DN && alt16 && lev == 1) ? tp1 : tp0, cur.tz) - blay;
        const bool cloudy = (cur.flag & 1u) != 0u;
        const bool first = (cur.flag & (DN ? 2u : 4u)) != 0u;
        const double cf = cur.cf;
        const bool anycld = __builtin_amdgcn_ballot_w64(cloudy) != 0ull;       // (wave-uniform)
        unsigned gbits = 0u;
        if constexpr (MODE == 4) gbits = (unsigned)((((unsigned long long)mlo | ((unsigned long long)mhi << 32)) >> (ig0 & 31)) & ((1ull << nvalid) - 1ull));
        fill_t(bin_tag, std::false_type{}, lev + dir, DN ? 0 : 1, cur);
        double qs[G], qsc[G];
#pragma unroll
        for (int k = 0; k < G; k++) {
            pk4 pk, pkt;
#pragma unroll
            for (int c = 0; c < NC; c++) if (c == slot) { pk = cc[k][c]; pkt = ct[k][NCT == 1 ? 0 : c]; }
            const scr4 ck = unpack4(pk), ckt = unpack4(pkt);
            float2 e[4], et[4];
#pragma unroll
            for (int jj = 0; jj < 4; jj++) e[jj] = RRLW_LUT_ENTRY(s_lut, code_index(ck.v[jj]));
            if (anycld) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++) et[jj] = RRLW_LUT_ENTRY(s_lut, min(code_index(ckt.v[jj]), (unsigned)NTBL));
            }
            {
                const pk4 nx = ld_c(sC, lev + NC * dir, k), nxt = ld_c(sCt, lev + NCT * dir, k);
#pragma unroll
                for (int c = 0; c < NC; c++) if (c == slot) { cc[k][c] = nx; ct[k][NCT == 1 ? 0 : c] = nxt; }
            }
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = 4 * k + jj;
                double atr, tfn;
                decode(ck.v[jj], e[jj], atr, tfn);
                double fr = row[j];
                if constexpr (BIN) fr = fma(fpl, row[16 + j] - fr, fr);
                const double bb = fr * fma(tfn, dpl, blay);
                // clear level: rtrnmr :617-627 / :705-716
                const double rad_s = fma(bb - rad[j], atr, rad[j]);
                const double updc = fma(bb - radc[j], atr, radc[j]);
                if (!anycld) {
                    rad[j] = rad_s;
                    radc[j] = seen ? updc : rad_s;
                    if constexpr (IDRV && !DN) {
                        const double tr = one_minus(atr);
                        drad[j] = times(drad[j], tr);
                        dradc[j] = seen ? times(dradc[j], tr) : drad[j];
                    }
                } else {
                    // cloudy level: rtrnmr :591-615 / :680-703, formed for every lane of the wave and selected
                    double atot, tftot;
                    decode(ckt.v[jj], et[jj], atot, tftot);
                    const double bbtot = fr * fma(tftot, dpl, blay);
                    const double gassrc = bb * atr;
                    double rad_c;
                    if constexpr (MODE == 2) {
                        double cr = cldrad[j], lr = clrrad[j], mr = radmr[j];
                        if (first) { cr = cf * rad[j]; lr = rad[j] - cr; mr = 0.0; }
                        const double ttot = one_minus(atot), tgas = one_minus(atr), clr = one_minus(cf);
                        const double cldsrc = bbtot * atot;
                        cr = fma(cr, ttot, cf * cldsrc);
                        lr = fma(lr, tgas, clr * gassrc);
                        rad_c = cr + lr;
                        const double radmod = fma(f1.y, cldsrc, fma(-f1.x, gassrc, mr * fma(f0.x, tgas, f0.y * ttot)));
                        const double oldcld = cr - radmod;
                        const double oldclr = lr + radmod;
                        mr = fma(-f2.y, oldcld, fma(f2.x, oldclr, -radmod));
                        // (the cloudy / clear parts and the carried correction of a lane that is clear at this level need not be kept:
                        // the next cloudy level of that column is the first of a block, istcld = 1, and sets them anew)
                        cldrad[j] = cr + mr;
                        clrrad[j] = lr - mr;
                        radmr[j] = mr;
                    } else {                        // rtrn :372-435 / rtrnmc: the same explicit fused form in both flavours (the array and the mask
                        double cfj = cf, efj = efcl;    // flavour of rtrnmc must round identically)
                        if constexpr (MODE == 4) { const bool on = (gbits >> j) & 1u; cfj = on ? 1.0 : 0.0; efj = on ? efcl : 0.0; }
                        if constexpr (MODE == 3) {
                            const float *c4 = reinterpret_cast<const float *>(&cf4[k]), *e4 = reinterpret_cast<const float *>(&ef4[k]);
                            cfj = c4[jj]; efj = e4[jj];
                        }
                        rad_c = fma(cfj, fma(bbtot, atot, -gassrc), fma(-rad[j], fma(efj, 1. - atr, atr), rad[j]) + gassrc);
                    }
                    rad[j] = cloudy ? rad_c : rad_s;
                    radc[j] = (cloudy || seen) ? updc : rad_s;
                    if constexpr (IDRV && !DN) {        // d/dT: rtrnmr :697-702 (cloudy), :712-714 (clear)
                        double cfd = cf;
                        if constexpr (MODE == 4) cfd = ((gbits >> j) & 1u) ? 1.0 : 0.0;
                        if constexpr (MODE == 3) cfd = reinterpret_cast<const float *>(&cf4[k])[jj];
                        const double tr = one_minus(atr);
                        const double d_s = times(drad[j], tr);
                        const double d_c = times(drad[j] * cfd, one_minus(atot)) + times(drad[j] * one_minus(cfd), tr);     // (rtrnmr :697-699: two products and their sum, rounded as written)
                        const double dc_s = seen ? times(dradc[j], tr) : d_s;
                        drad[j] = cloudy ? d_c : d_s;
                        dradc[j] = cloudy ? times(dradc[j], tr) : dc_s;
                    }
                }
            }
            qs[k] = (rad[4 * k] + rad[4 * k + 1]) + (rad[4 * k + 2] + rad[4 * k + 3]);
            qsc[k] = (radc[4 * k] + radc[4 * k + 1]) + (radc[4 * k + 2] + radc[4 * k + 3]);
        }
        if constexpr (DN) seen = seen || cloudy;
        if constexpr (G == 1) { red_put(slot, 0, qs[0]); red_put(slot, 1, qsc[0]); }
        else { red_put(slot, 0, qs[0] + qs[1]); red_put(slot, 1, qsc[0] + qsc[1]); }
        if constexpr (IDRV && !DN) {
            double ds = 0.0, dsc = 0.0;
#pragma unroll
            for (int k = 0; k < G; k++) {
                ds = ds + ((drad[4 * k] + drad[4 * k + 1]) + (drad[4 * k + 2] + drad[4 * k + 3]));
                dsc = dsc + ((dradc[4 * k] + dradc[4 * k + 1]) + (dradc[4 * k + 2] + dradc[4 * k + 3]));
            }
            red_put(slot, 2, ds);
            red_put(slot, 3, dsc);
        }
    };
    // The same level below the lowest cloud of the workgroup's blocks (layers 1 .. lbot - 1: clear in every column): the clear-sky update of
    // both streams - the arithmetic of the branch above that a wave without a cloudy lane takes, rtrnmr :617-627 / :705-716 - without the
    // cloudy level's inputs (total-optical-depth codes, cloud fraction, flag word, overlap factors / emissivity / mask words).
    auto level_lite = [&](auto bin_tag, auto dn_tag, int lev, int slot) __attribute__((always_inline)) {
        constexpr bool BIN = decltype(bin_tag)::value, DN = decltype(dn_tag)::value;
        constexpr int dir = DN ? -1 : 1;
        double fpl;
        const double *row = frac_row(lev, cur.w, fpl);
        const double blay = planck_at(tp0, tp0, cur.tl);
        const double dpl = planck_at(tp0, (DN && alt16 && lev == 1) ? tp1 : tp0, cur.tz) - blay;
        fill_t(bin_tag, std::true_type{}, lev + dir, DN ? 0 : 1, cur);
        double qs[G], qsc[G];
#pragma unroll
        for (int k = 0; k < G; k++) {
            pk4 pk;
#pragma unroll
            for (int c = 0; c < NC; c++) if (c == slot) pk = cc[k][c];
            const scr4 ck = unpack4(pk);
            float2 e[4];
#pragma unroll
            for (int jj = 0; jj < 4; jj++) e[jj] = RRLW_LUT_ENTRY(s_lut, code_index(ck.v[jj]));
            {
                const pk4 nx = ld_c(sC, lev + NC * dir, k);
#pragma unroll
                for (int c = 0; c < NC; c++) if (c == slot) cc[k][c] = nx;
            }
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = 4 * k + jj;
                double atr, tfn;
                decode(ck.v[jj], e[jj], atr, tfn);
                double fr = row[j];
                if constexpr (BIN) fr = fma(fpl, row[16 + j] - fr, fr);
                const double bb = fr * fma(tfn, dpl, blay);
                const double rad_s = fma(bb - rad[j], atr, rad[j]);
                const double updc = fma(bb - radc[j], atr, radc[j]);
                rad[j] = rad_s;
                radc[j] = seen ? updc : rad_s;
                if constexpr (IDRV && !DN) {
                    const double tr = one_minus(atr);
                    drad[j] = times(drad[j], tr);
                    dradc[j] = seen ? times(dradc[j], tr) : drad[j];
                }
            }
            qs[k] = (rad[4 * k] + rad[4 * k + 1]) + (rad[4 * k + 2] + rad[4 * k + 3]);
            qsc[k] = (radc[4 * k] + radc[4 * k + 1]) + (radc[4 * k + 2] + radc[4 * k + 3]);
        }
        if constexpr (G == 1) { red_put(slot, 0, qs[0]); red_put(slot, 1, qsc[0]); }
        else { red_put(slot, 0, qs[0] + qs[1]); red_put(slot, 1, qsc[0] + qsc[1]); }
        if constexpr (IDRV && !DN) {
            double ds = 0.0, dsc = 0.0;
#pragma unroll
            for (int k = 0; k < G; k++) {
                ds = ds + ((drad[4 * k] + drad[4 * k + 1]) + (drad[4 * k + 2] + drad[4 * k + 3]));
                dsc = dsc + ((dradc[4 * k] + dradc[4 * k + 1]) + (dradc[4 * k + 2] + dradc[4 * k + 3]));
            }
            red_put(slot, 2, ds);
            red_put(slot, 3, dsc);
        }
    };
    // `count` levels from layer `first` in the sweep direction; LITE: the clear-sky levels below the lowest cloud
    auto sweep = [&](auto bin_tag, auto dn_tag, auto lite_tag, int first, int count) __attribute__((always_inline)) {
        constexpr bool DN = decltype(dn_tag)::value, LITE = decltype(lite_tag)::value;
        constexpr int dir = DN ? -1 : 1;
        if (count <= 0) return;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (c == NC - 1) fill_t(bin_tag, lite_tag, first, DN ? 0 : 1, cur);
#pragma unroll
            for (int k = 0; k < G; k++) {
                cc[k][c] = ld_c(sC, first + c * dir, k);
                if constexpr (!LITE) { if (c < NCT) ct[k][c] = ld_c(sCt, first + c * dir, k); }
            }
        }
        auto one = [&](int lev, int slot) __attribute__((always_inline)) {
            if constexpr (LITE) level_lite(bin_tag, dn_tag, lev, slot); else level(bin_tag, dn_tag, lev, slot);
        };
        int n = 0;
        for (; n + NC <= count; n += NC) {
#pragma unroll
            for (int c = 0; c < NC; c++) one(first + (n + c) * dir, c);
            round_end(dn_tag, first + n * dir, NC);
        }
        const int rem = count - n;
#pragma unroll
        for (int c = 0; c < NC - 1; c++) {
            if (c < rem) one(first + (n + c) * dir, c);
        }
        if (rem > 0) round_end(dn_tag, first + n * dir, rem);
    };
    using std::true_type;
    using std::false_type;
    // the zone in two parts: layers lbot .. ltop with the cloudy-level machinery, layers 1 .. lbot - 1 below every cloud of the workgroup
    const int lbot = ltop >= 1 ? min(max(__builtin_amdgcn_readfirstlane(W.hbot[(blockIdx.x * nsb) / SORT_GROUP]), 1), ltop) : 1;

    // ------------------------------------------------------------------ downward: layers ltop .. lbot, then lbot - 1 .. 1
    if (any_bin) { sweep(true_type{}, true_type{}, false_type{}, ltop, ltop - lbot + 1); sweep(true_type{}, true_type{}, true_type{}, lbot - 1, lbot - 1); }
    else { sweep(false_type{}, true_type{}, false_type{}, ltop, ltop - lbot + 1); sweep(false_type{}, true_type{}, true_type{}, lbot - 1, lbot - 1); }
    // ------------------------------------------------------------------ surface: rtrnmr :629-652
    {
        const double reflect = 1. - a.emis[gc + (size_t)nct * (B - 1)];
        const double pb = W.percol[(size_t)(PC_PLANKBND + B - 1) * ncb + colc];
        const double dpb = IDRV ? W.percol[(size_t)(PC_DPLANKBND + B - 1) * ncb + colc] : 0.0;
        double fpl;
        const double *row = frac_row(1, any_bin ? bload_u32(sFw, off4) : 0u, fpl);
        double usum = 0.0, usumc = 0.0, dusum = 0.0;
#pragma unroll
        for (int j = 0; j < NG; j++) {
            const double fr = any_bin ? fma(fpl, row[16 + j] - row[j], row[j]) : row[j];
            const double rad0 = fr * pb;
            rad[j] = fma(reflect, rad[j], rad0);
            radc[j] = fma(reflect, radc[j], rad0);
            usum = usum + rad[j];
            usumc = usumc + radc[j];
            if constexpr (MODE == 2) { cldrad[j] = 0.0; clrrad[j] = 0.0; radmr[j] = 0.0; }
            if constexpr (IDRV) { drad[j] = times(fr, dpb); dradc[j] = drad[j]; dusum = dusum + drad[j]; }
        }
        red_put(0, 0, usum);
        red_put(0, 1, usumc);
        if constexpr (IDRV) { red_put(0, 2, dusum); red_put(0, 3, dusum); }
        round_end(false_type{}, 0, 1);              // level 0
    }
    seen = colcloud;
    // ------------------------------------------------------------------ upward: layers 1 .. lbot - 1, then lbot .. ltop
    if (any_bin) { sweep(true_type{}, false_type{}, true_type{}, 1, lbot - 1); sweep(true_type{}, false_type{}, false_type{}, lbot, ltop - lbot + 1); }
    else { sweep(false_type{}, false_type{}, true_type{}, 1, lbot - 1); sweep(false_type{}, false_type{}, false_type{}, lbot, ltop - lbot + 1); }
    if (incol) {                    // upward radiances at level ltop for k_sweepc<., 2>
        // (the hand-off address is formed again from the lane's column offset, hidden from common-subexpression elimination: held in
        // registers from the first use at the top of the kernel it cost the rtrnmr instantiations the two registers they were over budget)
        unsigned o4 = off4;
        asm volatile("" : "+v"(o4));
        double2 *hand_up = reinterpret_cast<double2 *>(W.hand) + ((size_t)quad * ncb + (o4 >> 2)) * 2;
#pragma unroll
        for (int k = 0; k < G; k++) {
            double2 *h = hand_up + (size_t)k * ncb * 2;
            h[hstream] = make_double2(rad[4 * k], rad[4 * k + 1]);        h[hstream + 1] = make_double2(rad[4 * k + 2], rad[4 * k + 3]);
            h[2 * hstream] = make_double2(radc[4 * k], radc[4 * k + 1]);  h[2 * hstream + 1] = make_double2(radc[4 * k + 2], radc[4 * k + 3]);
            if constexpr (IDRV) {
                h[3 * hstream] = make_double2(drad[4 * k], drad[4 * k + 1]);    h[3 * hstream + 1] = make_double2(drad[4 * k + 2], drad[4 * k + 3]);
                h[4 * hstream] = make_double2(dradc[4 * k], dradc[4 * k + 1]);  h[4 * hstream + 1] = make_double2(dradc[4 * k + 2], dradc[4 * k + 3]);
            }
        }
    }
}

#pragma clang fp contract(fast)
// ------------------------------------------------------------------------------------------------
// k_n1 : PROTOTYPE of the mapping BASELINE.json's north_star names, for cloud-free calls (icld = 0), kept to put a number beside the
//        production mapping ("north-star mapping"): ONE COLUMN PER WAVEFRONT, G-POINTS ACROSS LANES (three passes over the
//        152 padded g-slots), Planck / fraction tables in LDS, the transmittance either from the table in LDS or (EXPF) from
//        __builtin_amdgcn_exp2f, WAVE-REDUCE of the g-point radiances: the reduction covers all 16 bands at once, so the wave writes
//        the column's FINAL fluxes and heating rates - no per-band partial slabs, no k_flux / k_rates.  It reads k_layer's codes.
//        Per level the wave does what k_sweepc does once per 64 columns: 16 Planck pairs (one lane per band), the fraction rows, and a
//        64-lane reduction - which is why it loses (measured: profiles/round2_n1.md).
// ------------------------------------------------------------------------------------------------
constexpr int N1_WAVES = 8;                                   // columns per workgroup
constexpr int N1_PL_BYTES = 16 * 184 * 8, N1_FR_BYTES = 16 * 16 * 16 * 8;
constexpr int N1_WAVE_BYTES = 16 * 32 + 2 * 608;              // per wave: {blay, dpl, fpl, row} per band; up / down flux per level (<= 603 layers: see n1_max_nlay)
__host__ __device__ constexpr int n1_lds_bytes(int nlay) { return SWEEP_LUT_BYTES + N1_PL_BYTES + N1_FR_BYTES + N1_WAVES * (16 * 32 + 2 * 8 * (nlay + 1)); }

template <bool IDRV_UNUSED>
__global__ __launch_bounds__(64 * N1_WAVES) void k_n1(DevTables T, Workspace W, SweepArgs a, FluxOut out, const double *pz)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const float2 *s_lut = reinterpret_cast<const float2 *>(smem);
    double (*s_pl)[184] = reinterpret_cast<double (*)[184]>(smem + SWEEP_LUT_BYTES);                          // [band][181]
    double (*s_fr)[16][16] = reinterpret_cast<double (*)[16][16]>(smem + SWEEP_LUT_BYTES + N1_PL_BYTES);    // [band][row][g]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nlay = W.nlay, nct = a.nct;
    const size_t ncb = W.ncolb;
    unsigned char *wbase = smem + SWEEP_LUT_BYTES + N1_PL_BYTES + N1_FR_BYTES + (size_t)wv * (16 * 32 + 2 * 8 * (nlay + 1));
    double *s_pk = reinterpret_cast<double *>(wbase);                    // [band][4]: blay, dpl, fpl, (row index as double)
    double *s_dn = reinterpret_cast<double *>(wbase + 16 * 32), *s_up = s_dn + (nlay + 1);
    {   // stage: transmittance table, every band's Planck row and fraction rows
        const int tid = threadIdx.x, nth = 64 * N1_WAVES;
#ifndef RRLW_SWEEP_EXPF
        const double2 *src = reinterpret_cast<const double2 *>(T.stat + T.sl.lutf);
        double2 *dst = reinterpret_cast<double2 *>(smem);
        for (int i = tid; i < SWEEP_LUT_BYTES / 16; i += nth) dst[i] = src[i];
#endif
        for (int i = tid; i < 16 * 181; i += nth) s_pl[i / 181][i % 181] = T.stat[T.sl.totplnk + i];
        for (int i = tid; i < 16 * 16 * 16; i += nth) {
            const int b = i >> 8, r = (i >> 4) & 15, g = i & 15;
            const bool lo_bin = (LO_BINARY >> b) & 1u, up_bin = (UP_BINARY >> b) & 1u;
            const int ng = T.band[b].ng, fa = T.band[b].fracrefa, fb = T.band[b].fracrefb;
            const int na = lo_bin ? 9 : 1, nb = up_bin ? 5 : 1;
            double v = 0.0;
            if (g < ng) {
                if (r < 9) { if (r < na) v = T.ktab[fa + r * ng + g]; }
                else if (r < 14 && fb >= 0 && r - 9 < nb) v = T.ktab[fb + (r - 9) * ng + g];
            }
            s_fr[b][r][g] = v;
        }
        __syncthreads();
    }
    const int col = blockIdx.x * N1_WAVES + wv;
    if (col >= a.ncol) return;                      // (whole wave; no barrier follows)
    const size_t gc = (size_t)a.col0 + col;
    const int laytrop = W.laytrop[col];
    // lane -> g-slot of each pass: slot = 64 p + lane of the 152 padded slots (38 quads x 4)
    constexpr int NP1 = (4 * NQUAD + 63) / 64;       // passes over the padded g-slots (3 for 152, 4 for 256)
    int bnd[NP1], gi[NP1];
    double wt[NP1];
    size_t coff[NP1];
    bool ok[NP1];
#pragma unroll
    for (int p = 0; p < NP1; p++) {
        const int slot = 64 * p + lane, q = slot >> 2;
        int b = 0;
#pragma unroll
        for (int B = 1; B <= NBND; B++) if (q >= band_qstart(B)) b = B - 1;
        bnd[p] = b;
        gi[p] = 4 * (q - band_qstart(b + 1)) + (slot & 3);
        ok[p] = slot < 4 * NQUAD && gi[p] < T.band[b].ng;
        gi[p] = min(gi[p], 15);
        wt[p] = ok[p] ? 0.5 * T.delwave[b] : 0.0;
        coff[p] = ((size_t)min(q, NQUAD - 1) * nlay * ncb + col) * 4 + (slot & 3);        // + (lev - 1) * ncb * 4
    }
    const scr_t *codes = reinterpret_cast<const scr_t *>(W.scr[S_CODE]);      // (the prototype reads 32-bit codes: rrtmg_lw_hip_set_n1_prototype refuses other builds)
    // per level: lanes 0..15 (band = lane & 15) publish the band's Planck terms and fraction row
    const int mb = lane & 15;
    const bool m_lo_bin = (LO_BINARY >> mb) & 1u, m_up_bin = (UP_BINARY >> mb) & 1u;
    const int m_base_up = ((UP_ZERO >> mb) & 1u) ? 14 : (((UP_FROM_A >> mb) & 1u) ? 0 : 9);
    const int m_fwslot = fw_slot(mb + 1);
    auto publish = [&](int lev, int zlev, bool alt) {
        const double tl = a.tlay[gc + (size_t)nct * (lev - 1)], tz = a.tlev[gc + (size_t)nct * zlev];
        const bool lower = lev <= laytrop;
        unsigned w = 0x10000000u;
        if (lower ? m_lo_bin : m_up_bin) w = W.fw[((size_t)m_fwslot * nlay + (lev - 1)) * ncb + col];
        const int r0 = clampi((lower ? 0 : m_base_up) + (int)(w >> 28) - 1, 0, 14);
        const double fpl = (double)(w & 0x0fffffffu) * (1.0 / 268435456.0);
        const double *tp = s_pl[mb];
        const double blay = planck_at(tp, tp, tl);
        const double *tq = (alt && mb == 15) ? T.stat + T.sl.totplk16 : tp;        // (istart = 16 is not routed here; kept for symmetry)
        const double dpl = planck_at(tp, tq, tz) - blay;
        __builtin_amdgcn_wave_barrier();          // (the level before has read its entries)
        if (lane < 16) { s_pk[4 * mb] = blay; s_pk[4 * mb + 1] = dpl; s_pk[4 * mb + 2] = fpl; s_pk[4 * mb + 3] = (double)r0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto wave_sum = [&](double v) -> double {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    double rad[NP1];
#pragma unroll
    for (int p = 0; p < NP1; p++) rad[p] = 0.0;
    auto level = [&](int lev) -> double {
        double part = 0.0;
#pragma unroll
        for (int p = 0; p < NP1; p++) {
            const scr_t cj = ok[p] ? codes[coff[p] + (size_t)(lev - 1) * ncb * 4] : (scr_t)0;
            double atr, tfn;
            decode(cj, RRLW_LUT_ENTRY(s_lut, code_index(cj)), atr, tfn);
            const int b = bnd[p];
            const double blay = s_pk[4 * b], dpl = s_pk[4 * b + 1], fpl = s_pk[4 * b + 2];
            const int r0 = (int)s_pk[4 * b + 3];
            const double f0 = s_fr[b][r0][gi[p]], f1 = s_fr[b][r0 + 1][gi[p]];
            const double fr = f0 + fpl * (f1 - f0);
            const double bb = fr * (blay + tfn * dpl);
            rad[p] = rad[p] + (bb - rad[p]) * atr;
            part += wt[p] * rad[p];
        }
        return wave_sum(part);
    };
    // ---- downward
    if (lane == 0) s_dn[nlay] = 0.0;
    for (int lev = nlay; lev >= 1; lev--) {
        publish(lev, lev - 1, false);
        const double f = level(lev);
        if (lane == 0) s_dn[lev - 1] = f * T.fluxfac;
    }
    // ---- surface
    {
        publish(1, 0, false);          // fraction row of layer 1
        double part = 0.0;
#pragma unroll
        for (int p = 0; p < NP1; p++) {
            const int b = bnd[p];
            const double fpl = s_pk[4 * b + 2];
            const int r0 = (int)s_pk[4 * b + 3];
            const double f0 = s_fr[b][r0][gi[p]], f1 = s_fr[b][r0 + 1][gi[p]];
            const double fr = f0 + fpl * (f1 - f0);
            const double reflect = 1. - a.emis[gc + (size_t)nct * b];
            const double pb = W.percol[(size_t)(PC_PLANKBND + b) * ncb + col];
            rad[p] = fr * pb + reflect * rad[p];
            part += wt[p] * rad[p];
        }
        const double f = wave_sum(part);
        if (lane == 0) s_up[0] = f * T.fluxfac;
    }
    // ---- upward
    for (int lev = 1; lev <= nlay; lev++) {
        publish(lev, lev, false);
        const double f = level(lev);
        if (lane == 0) s_up[lev] = f * T.fluxfac;
    }
    // ---- outputs: lanes = levels
    for (int l0 = 0; l0 <= nlay; l0 += 64) {
        const int lev = l0 + lane;
        if (lev <= nlay) {
            const size_t o = gc + (size_t)nct * lev;
            const double u = s_up[lev], d = s_dn[lev];
            out.uflx[o] = u; out.dflx[o] = d; out.uflxc[o] = u; out.dflxc[o] = d;
            if (out.fnet) { out.fnet[o] = u - d; out.fnetc[o] = u - d; }
            if (lev < nlay) {
                const double h = T.heatfac * ((u - d) - (s_up[lev + 1] - s_dn[lev + 1])) / (pz[o] - pz[o + nct]);
                out.hr[o] = h; out.hrc[o] = h;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_flux  : sum of the group partials (rtrn :549-574), flux scaling (rtrn :580-594), net flux and heating rate (rtrn :583-604; output
//           copies src/rrtmg_lw_rad.nomcica.f90:563-583).  A workgroup covers a WINDOW of 256 columns and FLUX_LV levels and sums the
//           level above them once more, so that no second pass over the flux arrays is needed for the heating rate of its last layer.
//           The partials lie by POSITION, the caller's arrays by COLUMN (k_colsort): thread (t, .) sums the groups at position w0 + t
//           (coalesced), the values change places in LDS - position -> column; the same place where the columns are taken as they
//           lie - and thread (t, .) writes column w0 + t of the caller's arrays (coalesced).  (Gathered at the column's position,
//           eight bytes per lane, the slabs cost a third more - sixteen cache lines per access instead of four -, scattered to the
//           position's column the outputs 60 %; and the 256-column rows make this kernel 7 % faster than its 64-column predecessor
//           where no column moves at all.)
//           clear_from_total: a cloud-free call, the clear-sky stream equals the total-sky stream.  Only bands in [istart, iend] were
//           swept: the groups hold exactly those.
// ------------------------------------------------------------------------------------------------
constexpr int FLUX_LV = 8, FLUX_TY = 4;         // levels of a workgroup (+ 1: the level above, summed again), threads per column
static_assert(COLSORT_WIN * FLUX_TY <= 1024, "k_flux: a window x FLUX_TY threads");
constexpr int FLUX_LDS_BYTES = (FLUX_LV + 1) * 4 * COLSORT_WIN * 8;       // 72 KB: two workgroups per CU
template <bool IDRV>
__global__ __launch_bounds__(COLSORT_WIN * FLUX_TY) void k_flux(DevTables T, Workspace W, FluxOut out, const double *pz, int ncol, int col0, int nct,
                                                                int clear_from_total, int ngroups, unsigned long long gsz)
{   // gsz: slabs per group, a nibble each (1, or - SweepArgs::split - the group's bands: added first, in their order, like the group's workgroup does)
    constexpr int NV = 4, NL = (FLUX_LV + FLUX_TY) / FLUX_TY;      // values that change places at a time; levels per thread
    extern __shared__ __align__(16) unsigned char smem_f[];
    double (*s_v)[NV][COLSORT_WIN] = reinterpret_cast<double (*)[NV][COLSORT_WIN]>(smem_f);      // [FLUX_LV + 1]
    double kdu[IDRV ? NL : 1], kduc[IDRV ? NL : 1];         // d(flux)/dT of this thread's levels: they change places after the fluxes, in the same buffer
    const int t = threadIdx.x, ty = threadIdx.y, w0 = blockIdx.x * COLSORT_WIN;
    const int slot = w0 + t, col = w0 + t;                  // the position whose partials this thread sums, the column it writes
    const bool son = slot < ncol, con = col < ncol;
    const int nlay = W.nlay, lev0 = blockIdx.y * FLUX_LV;
    const size_t ncb = W.ncolb, gc = (size_t)col0 + col;
    const int dst = son ? pcol(W, slot) - w0 : t;           // where the position's column lies in the window
    // Partials arrive summed per group of bands.  Downward at and above the hand-off level of the position's 64-column block, and everywhere
    // in a cloud-free call, the clear-sky stream equals the total one and one value was written (k_sweepc); below, and upward, two.
    const int ltop = (son && !clear_from_total) ? W.hblk[slot >> 6] : 0;       // (uniform over the wave)
    const bool up1 = clear_from_total != 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {          // (unrolled: kdu / kduc stay in registers)
        const int lc = ty + k * FLUX_TY, lev = lev0 + lc;
        if constexpr (IDRV) { kdu[k] = 0.0; kduc[k] = 0.0; }
        if (lc <= FLUX_LV && son && lev <= nlay) {
            const bool dn1 = clear_from_total || lev >= ltop;
            double u = 0.0, d = 0.0, uc = 0.0, dc = 0.0, du = 0.0, duc = 0.0;
            const size_t pcb = (size_t)W.pcb;
            int slab = 0;
            for (int g = 0; g < ngroups; g++) {
                const int n = (int)((gsz >> (4 * g)) & 15ull);
                Part2 su{0.0, 0.0}, sd{0.0, 0.0}, sq{0.0, 0.0};
                for (int q = 0; q < n; q++, slab++) {
                    const size_t go = ((size_t)slab * (nlay + 1) + lev) * pcb + slot;
                    Part2 tu, td, tq{0.0, 0.0};
                    if (dn1) { td.a = W.gdn1[go]; td.b = td.a; } else td = W.gdn[go];
                    if (up1) { tu.a = W.gup1[go]; tu.b = tu.a; } else tu = W.gup[go];
                    if constexpr (IDRV) tq = W.gdp[go];
                    if (q == 0) { su = tu; sd = td; sq = tq; }
                    else { su.a = su.a + tu.a; su.b = su.b + tu.b; sd.a = sd.a + td.a; sd.b = sd.b + td.b; sq.a = sq.a + tq.a; sq.b = sq.b + tq.b; }
                }
                u = u + su.a; uc = uc + su.b;
                d = d + sd.a; dc = dc + sd.b;
                du = du + sq.a; duc = duc + sq.b;
            }
            u = u * T.fluxfac; d = d * T.fluxfac;
            if (clear_from_total) { uc = u; dc = d; duc = du; }
            else { uc = uc * T.fluxfac; dc = dc * T.fluxfac; }
            s_v[lc][0][dst] = u; s_v[lc][1][dst] = d; s_v[lc][2][dst] = uc; s_v[lc][3][dst] = dc;
            if constexpr (IDRV) { kdu[k] = du; kduc[k] = duc; }
        }
    }
    __syncthreads();
    for (int lc = ty; lc < FLUX_LV; lc += FLUX_TY) {
        const int lev = lev0 + lc;
        if (con && lev <= nlay) {
            const double u = s_v[lc][0][t], d = s_v[lc][1][t], uc = s_v[lc][2][t], dc = s_v[lc][3][t];
            const size_t o = gc + (size_t)nct * lev;
            out.uflx[o] = u; out.dflx[o] = d; out.uflxc[o] = uc; out.dflxc[o] = dc;
            if (out.fnet) { out.fnet[o] = u - d; out.fnetc[o] = uc - dc; }
            if (lev < nlay) {       // net flux and heating rate of the layer above the level
                const double net = u - d, netc = uc - dc;
                const double above = s_v[lc + 1][0][t] - s_v[lc + 1][1][t], abovec = s_v[lc + 1][2][t] - s_v[lc + 1][3][t];
                const double dp = pz[o] - pz[o + (size_t)nct];
                out.hr[o] = T.heatfac * (net - above) / dp;
                out.hrc[o] = T.heatfac * (netc - abovec) / dp;
            }
        }
    }
    if constexpr (IDRV) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int lc = ty + k * FLUX_TY, lev = lev0 + lc;
            if (lc < FLUX_LV && son && lev <= nlay) { s_v[lc][0][dst] = kdu[k]; s_v[lc][1][dst] = kduc[k]; }
        }
        __syncthreads();
        for (int lc = ty; lc < FLUX_LV; lc += FLUX_TY) {
            const int lev = lev0 + lc;
            if (con && lev <= nlay) {
                const size_t o = gc + (size_t)nct * lev;
                out.duflx_dt[o] = s_v[lc][0][t]; out.duflxc_dt[o] = s_v[lc][1][t];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_calibrate : reads n 16-byte words per array with 16 B per lane and writes them back (known byte counts),
//               launched by rrtmg_lw_hip_calibrate_stream so that a PMC run can check the unit and scale of
//               FETCH_SIZE / WRITE_SIZE on this chip (MI355X_MICROARCH.md, HBM section) in the same session.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_calibrate(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { double2 v = in[i]; v.x += 1.0; out[i] = v; }
}

// ------------------------------------------------------------------------------------------------
// k_fill_rows : the host-pointer entries do not copy a row of an input array whose values are all the same for the columns of a batch
//               (well-mixed gases, cloud and aerosol arrays outside the cloudy / dusty layers): entry blockIdx.y of the table (pinned host
//               memory, written by the entry's host threads) names a run of such rows and the 8-byte pattern they hold.
// ------------------------------------------------------------------------------------------------
struct RowFill { unsigned long long *dst; unsigned long long n, bits; };
constexpr int FILL_BLOCKS = 64;
constexpr unsigned long long FILL_MAX = 1ull << 20;      // values per table entry (a longer run is several entries)
__global__ __launch_bounds__(256) void k_fill_rows(const RowFill *__restrict__ tab)
{
    const RowFill e = tab[blockIdx.y];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < e.n; i += (unsigned long long)FILL_BLOCKS * 256) e.dst[i] = e.bits;
}

}  // namespace rrlw
