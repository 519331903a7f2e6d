/*
 * rrtmg_lw_hip.h - C ABI of the MI355X-native RRTMG_LW column solver (librrtmg_lw_hip.so).
 *
 * Drop-in boundary for the reference's GCM entry points.  A host model keeps calling
 *     use rrtmg_lw_init, only: rrtmg_lw_ini      (reference: src/rrtmg_lw_init.f90:47)
 *     use rrtmg_lw_rad,  only: rrtmg_lw          (reference: src/rrtmg_lw_rad.nomcica.f90:99-108,
 *                                                 McICA flavour src/rrtmg_lw_rad.f90:99-108)
 * through the ISO_C_BINDING shim modules shipped in rrtmg_lw_amd/fortran/, which forward to the
 * functions below.  Plain pointers and sizes only; every array is float64 in Fortran (column-major)
 * order exactly as the reference declares it:
 *     play,tlay,*vmr,cldfr,cicewp,cliqwp,reice,reliq  (ncol,nlay)      plev,tlev  (ncol,nlay+1)
 *     tsfc (ncol)   emis (ncol,16)   taucld (16,ncol,nlay)   tauaer (ncol,nlay,16)
 *     uflx,dflx,uflxc,dflxc,duflx_dt,duflxc_dt (ncol,nlay+1)   hr,hrc (ncol,nlay)
 *     McICA: cldfmcl,taucmcl,ciwpmcl,clwpmcl (140,ncol,nlay)   reicmcl,relqmcl (ncol,nlay)
 * Layer 1 is the surface layer; pressures in hPa.  The caller owns every array.
 *
 * Return value: 0 on success; non-zero = error, text available from rrtmg_lw_hip_last_error()
 * (the reference `stop 'MESSAGE'`s instead - src/rrtmg_lw_cldprop.f90:212,217,228,244,272; the
 * Fortran shim turns a non-zero code into `error stop` with the same text).
 * There is no CPU fallback: every entry fails with RRTMG_LW_HIP_ENODEVICE when no GPU is usable.
 */
#ifndef RRTMG_LW_HIP_H
#define RRTMG_LW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RRTMG_LW_HIP_OK 0
#define RRTMG_LW_HIP_EPHYSICS 1   /* the reference would `stop` (bad particle size / flag) */
#define RRTMG_LW_HIP_EARG 2
#define RRTMG_LW_HIP_ENODEVICE 3
#define RRTMG_LW_HIP_ENOTINIT 4
#define RRTMG_LW_HIP_EDATA 5      /* table / k-data file missing or malformed */
#define RRTMG_LW_HIP_EHIP 6       /* HIP runtime error */

#define RRTMG_LW_NBND 16
#define RRTMG_LW_NGPT 140

/* rrtmg_lw_ini(cpdair)  -  reference: src/rrtmg_lw_init.f90:47-194.
 * Reads the static tables (lw_static.bin) and the absorption-coefficient file (RRLWBLOB k-data,
 * original 16-g form; see rrtmg_lw_amd/kdata.py for the converters from rrtmg_lw_k_g.f90 and
 * rrtmg_lw.nc), performs the 256->140 g-point reduction and the LUT build on the host, and uploads the
 * packed, g-point-fastest tables to the device `device` (HIP ordinal of this process's GPU).
 * cpdair in J kg-1 K-1 sets heatfac (src/rrtmg_lw_init.f90:298). */
int rrtmg_lw_hip_init(const char *static_tables_path, const char *kdata_path, double cpdair, int device);

/* The same for SEVERAL GPUs driven by one process (SURVEY.md 8b `ndev`; the reference's calling model is one serial `do iplon` loop,
 * src/rrtmg_lw_rad.nomcica.f90:472: a non-MPI host has one process).  devices[0 .. ndev-1] are HIP ordinals; the host-pointer entries
 * rrtmg_lw_hip_run_nomcica and rrtmg_lw_hip_run_mcica then split their columns into ndev contiguous blocks and feed every device from
 * its own host thread (its own PCIe link, copy streams and workspace); every other entry works on devices[0].  An ordinal may be
 * listed more than once ("virtual devices": separate workspaces and streams on one GPU - how the fan-out is tested on a one-GPU box).
 * The Fortran shim calls this from rrtmg_lw_ini when RRTMG_LW_NDEV is set (devices RRTMG_LW_DEVICE, +1, ...; RRTMG_LW_VIRTUAL_DEVICES=1
 * keeps them all on RRTMG_LW_DEVICE). */
int rrtmg_lw_hip_init_devices(const char *static_tables_path, const char *kdata_path, double cpdair, int ndev, const int *devices);
int rrtmg_lw_hip_num_devices(void);     /* 0 before initialisation */

/* 1 if the loaded k-data is the synthetic stand-in (fluxes not physical), 0 if real, -1 if not initialised */
int rrtmg_lw_hip_kdata_is_standin(void);

void rrtmg_lw_hip_finalize(void);
/* Text of the calling THREAD's last error (concurrent callers each read their own); of the library's last one if this thread has had none. */
const char *rrtmg_lw_hip_last_error(void);

/* rrtmg_lw, non-McICA  -  reference: src/rrtmg_lw_rad.nomcica.f90:99-588.  HOST pointers.
 * icld is in/out (reset to 2 when outside [0,3], :456).  duflx_dt/duflxc_dt may be NULL unless idrv==1. */
int rrtmg_lw_hip_run_nomcica(
    int ncol, int nlay, int *icld, int idrv,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt);

/* The device-pointer entries below run on the state of the device their arrays live on (hipPointerGetAttributes on `play`): after
 * rrtmg_lw_hip_init_devices a one-process host with device-resident data on several GPUs uses all of them through the same entries;
 * arrays on a device the library was not initialised for are an error.  (The Mersenne-Twister generator, irng = 1 - one stream over all
 * columns of a call - lives on the first device.)  rrtmg_lw_hip_last_device_state: the index of the state this thread's last such call
 * ran on.  rrtmg_lw_hip_check(stream) waits for the stream and reports the first physics error of any state. */
int rrtmg_lw_hip_last_device_state(void);
/* Same contract with DEVICE pointers (a GPU-resident host model, and the benchmark's timed region).
 * Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) and the call
 * returns without synchronising; physics errors are reported by rrtmg_lw_hip_check(stream). */
int rrtmg_lw_hip_run_nomcica_device(
    int ncol, int nlay, int *icld, int idrv,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt, void *stream);

/* Synchronise `stream` and return the physics-error status of the work enqueued so far. */
int rrtmg_lw_hip_check(void *stream);

/* Prepared-column entry (HOST pointers): the physics sequence of the reference's column driver,
 * cldprop -> setcoef -> taumol -> rtrn|rtrnmr (src/rrtmg_lw.1col.f90:497-580), for `ncol` independent
 * columns that all have `nlayers` layers.  This is the interface the golden OUTPUT_RRTM files pin
 * (column amounts come from the input file, not from inatm's hydrostatic estimate).
 * Arrays are column-fastest: pavel,tavel,coldry,wbrodl,cldfrac,ciwp,clwp,rei,rel (ncol,nlayers);
 * pz,tz (ncol,0:nlayers); tbound,pwvcm (ncol); semiss (ncol,16); wkl (ncol,7,nlayers); wx (ncol,4,nlayers);
 * tauc (ncol,16,nlayers); taua (ncol,nlayers,16); outputs (ncol,0:nlayers).
 * istart..iend select the bands (1..16), any range (the reference's own sweeps support one band, iout > 0, or a range from band 1:
 * with iout = 0 their g-point counter starts at 1 whatever istart is, src/rrtmg_lw_rtrn.f90:354-360); istart==16 switches band 16 to the
 * 2600-3250 cm-1 Planck table
 * exactly as setcoef does (src/rrtmg_lw_setcoef.f90:233-252).  icld==1 -> rtrn, otherwise rtrnmr. */
int rrtmg_lw_hip_run_columns(
    int ncol, int nlayers, int istart, int iend, int icld, int idrv,
    const double *pavel, const double *tavel, const double *pz, const double *tz, const double *tbound,
    const double *semiss, const double *coldry, const double *wkl, const double *wbrodl, const double *wx,
    const double *pwvcm, int inflag, int iceflag, int liqflag, const double *cldfrac, const double *tauc,
    const double *ciwp, const double *clwp, const double *rei, const double *rel, const double *taua,
    double *totuflux, double *totdflux, double *fnet, double *htr,
    double *totuclfl, double *totdclfl, double *fnetc, double *htrc,
    double *dtotuflux_dt, double *dtotuclfl_dt);

/* McICA flavour of the prepared-column entry (HOST pointers): one Monte-Carlo sample of the column driver with imca = 1,
 * cldprmc -> setcoef -> taumol -> rtrnmc (src/rrtmg_lw.1col.f90:471-580), for `ncol` (column, sample) pairs.  Arrays as for
 * rrtmg_lw_hip_run_columns, the cloud arguments replaced by the sub-columns cldfmc, taucmc, ciwpmc, clwpmc (140,ncol,nlayers)
 * and reicmc, relqmc (ncol,nlayers).  The driver's result is the mean over its nmca = 200 samples, sample `ims` being generated
 * with permuteseed = ims * 140 (src/mcica_subcol_gen_lw.1col.f90:248-251). */
int rrtmg_lw_hip_run_columns_mcica(
    int ncol, int nlayers, int istart, int iend, int icld, int idrv,
    const double *pavel, const double *tavel, const double *pz, const double *tz, const double *tbound,
    const double *semiss, const double *coldry, const double *wkl, const double *wbrodl, const double *wx,
    const double *pwvcm, int inflag, int iceflag, int liqflag, const double *cldfmc, const double *taucmc,
    const double *ciwpmc, const double *clwpmc, const double *reicmc, const double *relqmc, const double *taua,
    double *totuflux, double *totdflux, double *fnet, double *htr,
    double *totuclfl, double *totdclfl, double *fnetc, double *htrc,
    double *dtotuflux_dt, double *dtotuclfl_dt);

/* McICA flavour ---------------------------------------------------------------------------------- */
/* rrtmg_lw, McICA  -  reference: src/rrtmg_lw_rad.f90:99-594 (cldprmc src/rrtmg_lw_cldprmc.f90:49, rtrnmc
 * src/rrtmg_lw_rtrnmc.f90:51).  HOST pointers; same contract as rrtmg_lw_hip_run_nomcica with the cloud
 * arguments replaced by the sub-column arrays cldfmcl,taucmcl,ciwpmcl,clwpmcl (140,ncol,nlay) and
 * reicmcl,relqmcl (ncol,nlay).  icld is reset to 2 when outside [0,3] (:469); icld==0 ignores the cloud arrays. */
int rrtmg_lw_hip_run_mcica(
    int ncol, int nlay, int *icld, int idrv,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfmcl, const double *taucmcl, const double *ciwpmcl, const double *clwpmcl,
    const double *reicmcl, const double *relqmcl, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt);

/* DEVICE-pointer variant (enqueue on `stream`, no synchronisation; see rrtmg_lw_hip_check). */
int rrtmg_lw_hip_run_mcica_device(
    int ncol, int nlay, int *icld, int idrv,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfmcl, const double *taucmcl, const double *ciwpmcl, const double *clwpmcl,
    const double *reicmcl, const double *relqmcl, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt, void *stream);

/* get_alpha  -  reference: src/mcica_subcol_gen_lw.f90:68-180.  HOST pointers; dz,cldfrac,alpha (ncol,nlay),
 * lat (ncol).  alpha is written only for icld = 4 or 5 (exponential / exponential-random overlap). */
int rrtmg_lw_hip_get_alpha(int ncol, int nlay, int icld, int idcor, double decorr_con, const double *dz, const double *lat,
                           int juldat, const double *cldfrac, double *alpha);

/* mcica_subcol_lw  -  reference: src/mcica_subcol_gen_lw.f90:183-291 (generate_stochastic_clouds :295-703,
 * kissvec :711-745, Mersenne Twister src/mcica_random_numbers.f90).  HOST pointers.
 * play,cldfrac,ciwp,clwp,rei,rel,alpha (ncol,nlay); tauc (16,ncol,nlay); outputs cldfmcl,ciwpmcl,clwpmcl,taucmcl
 * (140,ncol,nlay), reicmcl,relqmcl (ncol,nlay).  icld 1..5 = random, maximum-random, maximum, exponential,
 * exponential-random overlap (alpha may be NULL unless icld is 4 or 5); icld 0 returns without touching the
 * outputs.  irng is in/out (any non-zero value becomes 1): 0 = kissvec, one stream per column seeded from
 * its four lowest layer pressures; 1 = one Mersenne-Twister stream seeded with permuteseed, consumed in
 * (sub-column, column, layer) order.  kissvec: 4 <= nlay <= 603 (modules/parrrtm.f90:31; the column driver's mxlay is 203); a permuteseed that differs
 * from the previous call's rebuilds a small jump-ahead table on the host (one device synchronisation). */
int rrtmg_lw_hip_mcica_subcol(
    int ncol, int nlay, int icld, int permuteseed, int *irng, const double *play, const double *cldfrac, const double *ciwp,
    const double *clwp, const double *rei, const double *rel, const double *tauc, const double *alpha,
    double *cldfmcl, double *ciwpmcl, double *clwpmcl, double *reicmcl, double *relqmcl, double *taucmcl);

/* DEVICE-pointer variant (enqueue on `stream`; the irng = 1 stream is drawn on the device as well: chunk-parallel by jump-ahead of
 * MT19937, the chunk states cached per (permuteseed, ncol * nlay), a new seed costs one device synchronisation and 10-30 ms). */
int rrtmg_lw_hip_mcica_subcol_device(
    int ncol, int nlay, int icld, int permuteseed, int *irng, const double *play, const double *cldfrac, const double *ciwp,
    const double *clwp, const double *rei, const double *rel, const double *tauc, const double *alpha,
    double *cldfmcl, double *ciwpmcl, double *clwpmcl, double *reicmcl, double *relqmcl, double *taucmcl, void *stream);

/* mcica_subcol_lw + McICA rrtmg_lw in one call (the sequence a host model runs every radiation step,
 * README.md / src/rrtmg_lw_rad.f90:140-150), without materialising the (140,ncol,nlay) arrays: the generator
 * leaves a 140-bit cloud mask per (column, layer) on the device and cldprmc/rtrnmc consume it directly.
 * Results equal rrtmg_lw_hip_mcica_subcol followed by rrtmg_lw_hip_run_mcica.  Cloud arguments are the
 * generator's inputs (cldfr,cicewp,cliqwp,reice,reliq,alpha (ncol,nlay); taucld (16,ncol,nlay)); icld in
 * [0,5] selects the overlap for the generator and comes back as rrtmg_lw would leave it. */
int rrtmg_lw_hip_run_mcica_subcol(
    int ncol, int nlay, int *icld, int idrv, int permuteseed, int *irng,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *alpha, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt);

int rrtmg_lw_hip_run_mcica_subcol_device(
    int ncol, int nlay, int *icld, int idrv, int permuteseed, int *irng,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *alpha, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt, void *stream);

/* Tuning / introspection ------------------------------------------------------------------------- */
/* Aggregation of small calls.  A host model that calls rrtmg_lw per chunk of a few dozen columns (the reference is called that way:
 * `do iplon = 1, ncol` inside, pcols-sized chunks outside, src/rrtmg_lw_rad.nomcica.f90:472) pays ~0.5 ms of launch latency per call on
 * the GPU whatever the chunk size.  queue_begin fixes what the chunks share (layer count, flags); queue_add takes one chunk with
 * exactly the argument list of rrtmg_lw_hip_run_nomcica - it only records the pointers, the arrays must stay valid and unchanged until
 * the flush; queue_flush packs all queued chunks column-wise into one pinned staging set, runs ONE pass over all their columns and
 * writes every chunk's outputs (and its icld, reset as rrtmg_lw resets it) back.  Errors of the pass are reported by the flush. */
int rrtmg_lw_hip_queue_begin(int nlay, int icld, int idrv, int inflglw, int iceflglw, int liqflglw);
int rrtmg_lw_hip_queue_add(
    int ncol, int *icld,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt);
int rrtmg_lw_hip_queue_flush(void);
/* columns queued since the last flush */
int rrtmg_lw_hip_queue_columns(void);

/* What the library was built with: 0 for the shipped library, 8 for the 256-g-point one.  Bit 0: tuning build (-DRRLW_TUNE: only the
 * benchmark's kernels), bit 1: knock-out switch (WRONG results, timing only), bit 2: numerics variant (other code width / decode /
 * division than the product's), bit 3: -DRRLW_G256, bit 4: kernel-geometry switches (same results, other speed).
 * rrtmg_lw_hip_init[_devices] refuses a library with bit 1 or 2 unless RRTMG_LW_ALLOW_TUNE_BUILD=1 is set. */
unsigned rrtmg_lw_hip_build_flags(void);
/* Columns processed per internal batch (bounds the device workspace); 0 = back to the default, which follows the call's layers: 262144 at up
 * to 96 layers, 131072 at 137 (the next lower power of two of 262144 x 72 / nlay) - the benchmark's call (1e6 x 72 layers, rtrnmr) then holds
 * 36.9 GB of workspace, the 137-layer call with aerosol and d/dT 38.7 GB; set_batch(65536) brings the 72-layer call to 9.3 GB at 4 % of its
 * speed (profiles/round5_batch_sweep.md) (0.06-0.16 MB of device workspace per column at 72 layers, by call shape: see rrtmg_lw_hip_workspace_bytes). */
int rrtmg_lw_hip_set_batch(int ncol_batch);
/* on = 1: device-pointer entries run the sweeps / k_flux of column batch i on a second stream while k_layer of batch i+1 runs on the
 * caller's stream (second scratch set, +0.1 MB of workspace per column).  Default 0: a sweep workgroup owns a CU (transmittance table in
 * LDS, all vector registers), so the two do not share one (measured: 1-2 % faster on 1e6 cloudy columns).  on = 0 frees the second set at the next workspace
 * allocation. */
int rrtmg_lw_hip_set_overlap(int on);
/* CU partition of that overlap: k_layer of batch i+1 runs on `layer_cus` of the device's compute units (rounded to a multiple of 8, the
 * same number from every XCD: hipExtStreamCreateWithCUMask), the sweeps and k_flux of batch i on the others; the first batch's k_layer and
 * the last batch's sweeps, which have nothing beside them, take the whole chip.  layer_cus > 0 switches the overlap on; 0 removes the
 * partition.  Results do not depend on it.  rrtmg_lw_hip_cu_partition() returns the CUs of k_layer's share (0: none). */
int rrtmg_lw_hip_set_cu_partition(int layer_cus);
int rrtmg_lw_hip_cu_partition(void);
/* A cloudy batch too small to fill the chip is a latency chain of kernels, three of them sweeps (down above the clouds, the cloud zone, up
 * above the clouds).  Batches of up to `ncol` columns (default 4096; RRTMG_LW_ONE_SWEEP_MAX) take ONE sweep launch per band group instead:
 * the cloud-zone kernel walks all levels.  0 = never.  Results do not depend on it (bit for bit).  Returns the previous value. */
int rrtmg_lw_hip_set_one_sweep_max(int ncol);
/* ... and its sweeps put twelve waves on each of a few CUs: 1 024 columns are 16 blocks of 64, i.e. 16 workgroups per group of bands.
 * Batches of up to `ncol` columns (default 768: sixteen bands x twelve blocks still find a CU each; RRTMG_LW_SPLIT_MAX) are swept ONE band
 * per workgroup instead - one to four waves - and leave a flux partial per band; k_flux adds the bands of a group first, in the order the group's workgroup adds them,
 * so the results do not depend on it (bit for bit).  0 = never.  Returns the previous value. */
int rrtmg_lw_hip_set_split_max(int ncol);
/* A device-resident call of one batch that does not fill the chip is a chain of a dozen dependent launches on four streams.  The second
 * call with the same arguments (shape, flags, every array pointer - a host model hands over the same arrays step after step) is
 * captured as a graph, later ones are ONE hipGraphLaunch on the caller's stream (up to 8 graphs are kept).  Calls of up to `ncol`
 * columns are taken this way (default 16384; RRTMG_LW_GRAPH_MAX; 0 = never).  Results do not depend on it (bit for bit).  Returns the
 * previous value.  rrtmg_lw_hip_graph_stats: graphs captured / calls replayed since initialisation. */
int rrtmg_lw_hip_set_graph_max(int ncol);
void rrtmg_lw_hip_graph_stats(long long *captures, long long *replays);
/* k_layer stages, per workgroup of 256 columns of one layer, the reference-pressure planes of the absorption tables those columns
 * need.  Where the columns of a model level lie within one plane of each other (a grid that is nearly the same in every column) three
 * planes do; on a terrain-following grid (surface pressures of 550 .. 1040 hPa side by side: up to three planes apart,
 * reference src/rrtmg_lw_setcoef.f90:276-284) such workgroups are taken by a second launch that stages five planes and ten minor-gas
 * temperature slices (GCM entries; default on, RRTMG_LW_WIDE_WINDOW=0 to switch off: every workgroup then keeps the narrow window and
 * the cells outside it read the tables through the vector L1 - 1.6x the kernel's time on such a grid).  Results do not depend on it
 * (bit for bit).  Returns the previous value. */
int rrtmg_lw_hip_set_wide_window(int on);

/* k_layer takes the sixteen bands of a (window of 256 columns, layer) in ONE workgroup - a 100 us chain; a batch with too few such pairs to
 * fill the chip (up to ~2 000 columns of 72 layers) spreads them over two or four workgroups along the staging passes (default on,
 * RRTMG_LW_LAYER_SPLIT=0 to switch off).  Results do not depend on it (bit for bit).  Returns the previous value. */
int rrtmg_lw_hip_set_layer_split(int on);
/* The sweeps decide per wavefront - 64 consecutive columns of a batch - where the clouds end; one deep tower among 64 shallow columns sends
 * all of them through the cloud-zone sweep up to its top.  By default (RRTMG_LW_COLSORT=0 to switch off) the columns of a cloudy batch
 * - rtrn, rtrnmr and the McICA entries, where the key is the grid-mean cloud fraction the sub-columns are drawn from - are therefore
 * TAKEN in another order than they lie: within each window of 256 consecutive columns by their highest cloudy layer,
 * deepest first (k_colsort); the caller's arrays stay as they are and are read / written through that order.  A window is reordered only
 * where that takes at least `min_gain` block-levels out of the cloud zone (sum over its four 64-column blocks of the highest cloudy layer,
 * as the columns lie against sorted; default 24, RRTMG_LW_COLSORT_MIN; < 0 keeps the value): reading the caller's arrays out of order has a
 * price.  on = 1 / off = 0; results do not depend on it (bit for bit).  Returns the previous `on`. */
int rrtmg_lw_hip_set_column_sort(int on, int min_gain);
/* the threshold in force (min_gain above; values beyond 2^24 are taken as 2^24 = never) */
int rrtmg_lw_hip_column_sort_min(void);
/* Bytes of device memory the library holds right now, over all its devices: per-batch workspace (it holds what the call shapes seen so far need and only
 * grows: per column of the batch at 72 layers 62 KB for cloud-free calls, 148 KB for rtrnmr, 152 KB for rtrn, 166 KB with idrv = 1; 310 KB at
 * 137 layers with idrv = 1: rrtmg_lw_hip_set_batch trades it against launch count), host-entry staging, McICA masks, the slab buffer
 * (<= 512 MB, or one slab) and the cached chunk states (<= 2 x 180 MB) of the Mersenne-Twister stream. */
long long rrtmg_lw_hip_workspace_bytes(void);
/* Measurement only: 1 = cloud-free GCM calls take the prototype of the "one column per wavefront" mapping (k_n1, profiles/round2_n1_expf.md)
 * instead of the production sweeps.  Off after every rrtmg_lw_hip_init; rrtmg_lw_hip_n1_prototype() reports it. */
int rrtmg_lw_hip_set_n1_prototype(int on);
int rrtmg_lw_hip_n1_prototype(void);
/* Number of g-point chunks the sweep kernel distributes over threads (one partial flux slab each). */
int rrtmg_lw_hip_num_chunks(void);
/* g-points of this build: 140 (librrtmg_lw_hip.so, the reference's shipped model) or 256 (librrtmg_lw_hip_g256.so: every band keeps its 16
 * original g-points - the accuracy mode the reference keeps commented out in modules/parrrtm.f90:40-41,77-110; same symbols; the McICA
 * sub-column arrays and masks then hold 256 sub-columns; select it by linking / loading that library instead). */
int rrtmg_lw_hip_gpoints(void);

/* Per-kernel timing with HIP events recorded on the launch stream (used by bench.py's roofline leg).
 * profile_end synchronises the device and writes "<kernel> <launches> <total_ms>" lines into buf. */
void rrtmg_lw_hip_profile_begin(void);
int rrtmg_lw_hip_profile_end(char *buf, int len);

/* Optional: pin a host array that will be passed to the host-pointer entries again and again (hipHostRegister).  Their H2D / D2H
 * copies then run as asynchronous DMA overlapped with the kernels of the neighbouring column batches. */
int rrtmg_lw_hip_host_register(void *ptr, long long bytes);
int rrtmg_lw_hip_host_unregister(void *ptr);
/* 1 when [ptr, ptr + bytes) lies inside a range registered above - the only arrays the entries copy from where they lie; everything else
 * goes through the library's own pinned staging (an array is NOT pinned because its ends share pages with registered neighbours). */
int rrtmg_lw_hip_host_is_registered(const void *ptr, long long bytes);
/* Optional: declare [ptr, ptr + bytes) STATIC - its contents stay as they are until rrtmg_lw_hip_host_changed(ptr, .).  The host-pointer
 * entries look at every input row of every call for "one value for all columns of the batch" (such rows are filled on the device instead
 * of copied: well-mixed gases handed over as full arrays, zero aerosol and cloud rows); that scan reads ~14 KB per 72-layer column and bounds
 * the entries once the arrays are pinned.  For a static array the answer is kept per (array, column batch) and its rows are not read again.
 * An array that is not declared is scanned on every call (the reference's semantics: inputs may change between calls).
 * rrtmg_lw_hip_host_changed: the contents have changed - the cached scans are dropped; keep = 0 also withdraws the declaration (call it
 * before the array is freed).  Fortran: rrtmg_lw_static / rrtmg_lw_changed (module rrtmg_lw_init). */
int rrtmg_lw_hip_host_static(const void *ptr, long long bytes);
int rrtmg_lw_hip_host_changed(const void *ptr, int keep);
/* Concurrent callers of rrtmg_lw_hip_run_nomcica (an OpenMP host model calling per chunk of columns from several threads, the reference's
 * calling model: serial inside a call, src/rrtmg_lw_rad.nomcica.f90:472): a call of at most RRTMG_LW_COMBINE_MAX (default 8192) columns
 * that finds another call in flight is left in a list; whoever holds the turn solves everything that has gathered with the same
 * (nlay, icld, idrv, inflglw, iceflglw, liqflglw) as ONE device pass and wakes the owners.  Results do not depend on it; a physics error
 * is reported to the call whose columns caused it.  RRTMG_LW_COMBINE=0 restores the plain lock.  This returns the calls that came through
 * the combining entry and the device passes that served them, since the library was loaded. */
void rrtmg_lw_hip_combine_stats(long long *calls, long long *passes);

/* PMC calibration: one kernel that reads `bytes` and writes `bytes` with 16 B per lane (known HBM traffic), so that a
 * rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass can fix the counters' unit and scale in the same session. */
int rrtmg_lw_hip_calibrate_stream(long long bytes);

#ifdef __cplusplus
}
#endif
#endif
