// C-ABI implementation (include/rrtmg_lw_hip.h): device state, workspace, batching, kernel dispatch.
// Replaces the serial `do iplon = 1, ncol` loop of the reference's rrtmg_lw
// (src/rrtmg_lw_rad.nomcica.f90:472-586) with batched launches of the kernels in kernels.hip.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdarg>
#include <mutex>
#include <string>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <thread>
#include <vector>

#include "../../include/rrtmg_lw_hip.h"
#include "kernels.hip"
#include "mtjump.hpp"

namespace {

using namespace rrlw;

constexpr int BLOCK = 256;
// columns per batch of the host-pointer entries (H2D | kernels | D2H pipeline); RRTMG_LW_HOST_BATCH overrides (measurements)
const int HOST_BATCH = []() { const char *e = getenv("RRTMG_LW_HOST_BATCH"); const int v = e ? atoi(e) : 0; return v >= 64 ? v : 16384; }();     // 16 384: the entries are bound by the host threads, short batches fill and drain the pipeline sooner (524 288 columns pinned: 93.8 ms against 119.6 with 32 768)

// staging sets of the host-pointer pipeline (device staging, pinned host staging, events): the calling thread may be this many batches
// ahead of the batch whose outputs it unpacks.  With two, the thread's waits at the head of every iteration (the H2D of batch i - 2 out of
// the pinned set, the D2H of batch i - 2 into it) left the copy engines and the kernels taking turns: a trace of a 524 288-column call
// showed 11 ms of overlap between 44 ms of kernels and 52 ms of copies.
constexpr int HOST_SETS = 4;

struct KissTable { unsigned long long stride = 0; long long seed = -1; KissJump host[KJ_NGROUP + 1]; KissJump *dev = nullptr; };

// Columns per internal batch when the caller has not chosen (rrtmg_lw_hip_set_batch(0) comes back to this): 262 144 at up to 96 layers,
// the next lower power of two of 262 144 x 72 / nlay above (131 072 at 137 layers) - the workspace grows with the layers, the step
// time does not need it.  Measured (profiles/round5_batch_sweep.md, 1e6 columns, 32 768 / 65 536 / 131 072 / 262 144 per batch): cloudy
// 66.1 / 61.3 / 59.3 / 58.9 ms, clear 46.4 / 43.6 / 42.1 / 39.7, deep clouds 83.5 / 77.8 / 76.0 / 74.4; 137 layers with aerosol and
// d/dT (5e5 columns) 68.5 / 65.9 / 65.0 / 65.2.
constexpr int DEFAULT_BATCH = 262144;
struct State {
    bool init = false;
    int device = -1;
    HostTables H;
    DevTables D;
    double *d_ktab = nullptr, *d_stat = nullptr;
    // workspace
    Workspace W{};
    void *ws_base = nullptr;
    size_t ws_bytes = 0;
    int ws_nlay = 0, ws_ncolb = 0;
    bool ws_cloud = false, ws_mc = false, ws_gdp = false, ws_efcl = false, ws_ovl = false;
    int ws_groups = 0;
    size_t ws_slabcols = 0;      // capacity of each partial slab array (gdn1 ..) in columns x slabs: ws_groups wide slabs, or - a small batch whose bands
                                 // leave a slab each (split sweeps) - sixteen narrow ones
    int *d_err = nullptr;
    // per-column / cloud-property arrays exist twice: k_colprep + k_cloudscan / k_cloudlay of batch i+1 run on `aux` while batch i is in k_layer/k_sweep
    struct PrepSet { double *percol; int *laytrop, *ncbands; double *odcld, *efcl; int *cflag; int *btop, *order, *hgrp, *hblk, *bbot, *hbot; double2 *ovl; int *perm, *wsort; double *tlayc, *tlevc, *cldfc; int *wide; } prep[2] = {};
    // the per-cell scratch written by k_layer and read by the sweeps exists twice as well: sweeps / k_flux of batch i (HBM-bound)
    // run on `sw` while k_layer of batch i+1 (latency-bound) runs on the caller's stream
    struct ScrSet { unsigned *scr[NSCR]; unsigned *fw; } scrset[2] = {};
    hipStream_t aux = nullptr;
    // streams of the sweep launches: `main` (the split pipeline's sweep stream) and q[0..2], on which the launches of the 3-, 2- and 1-quad
    // bands run beside the 4-quad launch.  Set 0 may use the whole chip; set 1 exists with a CU partition (rrtmg_lw_hip_set_cu_partition)
    // and is confined to the sweeps' share of the CUs.
    struct SweepSet { hipStream_t main = nullptr, q[3] = {nullptr, nullptr, nullptr}; hipEvent_t go = nullptr, done[3] = {nullptr, nullptr, nullptr}; } swset[2];
    // k_layer's streams of the partitioned pipeline: lay_u the whole chip (first batch of a call: nothing runs beside it), lay_m its share
    hipStream_t lay_u = nullptr, lay_m = nullptr;
    int cu_layer = 0;            // CUs of k_layer's share (0: no partition), set by rrtmg_lw_hip_set_cu_partition
    int cu_total = 0;
    bool sweep_fanout = true;
    hipEvent_t ev_last = nullptr;           // end of the previous device-entry call (calls on different streams share the workspace)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;      // run_prep: k_colprep beside k_cloudscan in a single-batch call
    bool ev_last_valid = false;
    hipEvent_t ev_in = nullptr, ev_ready[2] = {nullptr, nullptr}, ev_layer[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    // McICA sub-column masks of all columns of the current call
    unsigned *mask = nullptr;
    size_t mask_bytes = 0;
    double *d_rnd = nullptr;   // one slab of Mersenne-Twister deviates (irng = 1)
    size_t rnd_bytes = 0;
    bool batch_set = false; // rrtmg_lw_hip_set_batch has named a size (otherwise: auto_batch(nlay))
    int batch = DEFAULT_BATCH;     // columns per internal batch: 0.06-0.16 MB of workspace per column at 72 layers by call shape (38 GB of the 288 for the benchmark's); measured per 1e6 cloudy columns (end of round 2): 131072: 63.1 ms, 262144: 62.7, 524288: 61.7-62.3, 1048576: 61.2 - within the run-to-run spread, not worth the memory
    bool split_sweep = false;    // run the sweeps / k_flux of batch i concurrently with k_layer of batch i+1 (device entries).  Off by default: a sweep
                                 // workgroup owns a CU (transmittance table in LDS), so the two do not share a CU (measured: 1-2 % gain for twice the code scratch)
    bool sweep_attrs = false;    // the sweeps' dynamic-LDS limit has been raised on this device
    bool n1 = false;             // rrtmg_lw_hip_set_n1_prototype(1): cloud-free calls take the north-star-mapping prototype k_n1 (measurement only)
    bool ws_two_scr = false;     // the workspace holds the second scratch set that split_sweep needs
    // host-entry staging
    void *stage_base = nullptr;
    size_t stage_bytes = 0;
    hipStream_t stream = nullptr;
    hipStream_t cp_in = nullptr, cp_out = nullptr;      // host-pointer entries: H2D and D2H copy streams
    double *h_tot = nullptr;                            // pinned host scratch of the non-McICA host entry: tauctot of HOST_SETS column batches
    size_t h_tot_doubles = 0;
    // pinned host staging of the host-pointer entries, HOST_SETS sets like the device staging: `in` = the rows of the caller's PAGEABLE input
    // arrays that travel (packed by the host threads, then one DMA per run of rows), `out` = the pageable output arrays' rows (DMA, then
    // unpacked by the host threads), `fill` = the table of k_fill_rows (rows that do not travel)
    struct HostSet { char *in = nullptr; size_t in_cap = 0; char *out = nullptr; size_t out_cap = 0; char *fill = nullptr; size_t fill_cap = 0; } hset[HOST_SETS];
    hipEvent_t ev_h2d[HOST_SETS] = {}, ev_cmp[HOST_SETS] = {}, ev_d2h[HOST_SETS] = {};
    KissTable kiss;             // the kissvec generator's jump table of the last (draws per sub-column, permuteseed) pair, on this device
    std::string err;
    // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
    bool profile = false;
    // small device-resident calls replayed as ONE graph (run_pipelined): what a call's launches depend on -> the instantiated graph
    struct GraphEnt { std::vector<unsigned char> key; hipGraphExec_t exec = nullptr; int seen = 0; bool failed = false; unsigned long long used = 0; };
    std::vector<GraphEnt> graphs;
    unsigned long long graph_clock = 0;
    hipStream_t cap = nullptr;         // the stream the launches of such a call are captured on (the caller's may be the null stream)
    long long graph_replays = 0, graph_captures = 0;
    struct ProfRec { const char *name; hipEvent_t a, b; };
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> evpool;
};
// One State per device the library drives (rrtmg_lw_hip_init: one; rrtmg_lw_hip_init_devices: several GPUs from one process, SURVEY.md 8b
// `ndev`).  Every function below works on the calling thread's CURRENT state `G`: state 0 for every entry, and inside the host-pointer
// entries each worker thread of the fan-out (nomcica_host) takes the state of the device it feeds.  The same physical device may appear
// in several states ("virtual devices": separate workspaces, streams and table copies - how the fan-out is tested on a one-GPU box).
constexpr int MAXDEV = 16;
State g_states[MAXDEV];
int g_ndev = 1;
thread_local State *g_cur = &g_states[0];
#define G (*g_cur)
std::mutex g_mu;

// aggregation of small calls (rrtmg_lw_hip_queue_*): recorded chunks and the pinned staging set they are packed into
struct QueuedChunk { int ncol; int *icld; const double *in[24]; double *out[8]; };       // in[23]: alpha of the generator (fused McICA calls; optional)
struct ChunkQueue {
    bool open = false;
    int nlay = 0, icld = 0, idrv = 0, inflg = 0, iceflg = 0, liqflg = 0;
    long long ncol = 0;
    std::vector<QueuedChunk> chunks;
    double *pinned = nullptr;
    size_t pinned_doubles = 0;
} Q;

// The HIP current device is per host thread: an entry called from another thread (an OpenMP host model, a Python worker) must
// select the device the workspace lives on, and leaves the caller's device selection as it found it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    DeviceGuard()
    {
        if (G.init && hipGetDevice(&prev) == hipSuccess && prev != G.device) switched = hipSetDevice(G.device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define ENTRY_LOCK std::lock_guard<std::mutex> lk(g_mu); DeviceGuard dg_

// Device-pointer entries: the state (rrtmg_lw_hip_init_devices) of the device the caller's arrays live on, found from the first array
// (hipPointerGetAttributes) - a one-process host with device-resident data on several GPUs uses all of them through the same entries.
// Several states on ONE device (the virtual devices of the tests) take such calls in turn.  A pointer the runtime does not know keeps
// the first state (the behaviour until round 5); arrays on a device the library was not initialised for are an error.  g_cur is
// thread-local: put back when the entry returns.
thread_local int tl_dev_state = 0;          // the state this thread's last device-pointer entry ran on (rrtmg_lw_hip_last_device_state)
struct StateSelect {
    State *keep;
    bool bad = false;
    int dev = -1;
    // first_only: work that lives on the first state whatever the arrays' device (the Mersenne-Twister generator: ONE stream over all columns
    // of a call, its jump-ahead tables on the first device) - arrays elsewhere are an error
    explicit StateSelect(const void *p, bool first_only = false) : keep(g_cur)
    {
        static unsigned turn = 0;
        tl_dev_state = 0;
        g_cur = &g_states[0];
        if (!p || g_ndev <= 1) return;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return; }
        dev = at.device;
        if (first_only) { bad = g_states[0].device != dev; return; }
        int match[16], n = 0;
        for (int d = 0; d < g_ndev && n < 16; d++) if (g_states[d].init && g_states[d].device == dev) match[n++] = d;
        if (n == 0) { bad = true; return; }
        tl_dev_state = match[turn++ % (unsigned)n];
        g_cur = &g_states[tl_dev_state];
    }
    ~StateSelect() { g_cur = keep; }
};
#define ENTRY_LOCK_FOR(ptr, ...) std::lock_guard<std::mutex> lk(g_mu); StateSelect ss_(ptr, ##__VA_ARGS__); DeviceGuard dg_; \
    if (ss_.bad) return fail(RRTMG_LW_HIP_EARG, "the arrays live on device %d, which the library was not initialised for (rrtmg_lw_hip_init_devices)", ss_.dev)

// The text of an error belongs to the THREAD whose call failed (concurrent callers: another thread's failure a moment later must not
// replace it before the caller has read it); the state keeps a copy for threads that have had no error of their own.
thread_local std::string tl_err;
// the library's latest error text whatever the thread, for a thread that has none of its own (rrtmg_lw_hip_last_error): under its own
// small lock, never the entry lock - a caller polling for errors must not wait out another thread's solve
std::mutex g_lasterr_mu;
std::string g_lasterr;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    G.err = buf;
    tl_err = buf;
    { std::lock_guard<std::mutex> lk(g_lasterr_mu); g_lasterr = buf; }
    return code;
}

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail(RRTMG_LW_HIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

const char *physics_message(int code)
{
    switch (code) {   // texts of the reference's `stop` statements, src/rrtmg_lw_cldprop.f90:212,217,228,244,272
    case E_ICE_SMALL: return "ICE RADIUS TOO SMALL";
    case E_ICE_BOUNDS: return "ICE RADIUS OUT OF BOUNDS";
    case E_ICE_GEN_BOUNDS: return "ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS";
    case E_LIQ_BOUNDS: return "LIQUID EFFECTIVE RADIUS OUT OF BOUNDS";
    case E_BAD_FLAG: return "INVALID CLOUD PROPERTY FLAG";
    case E_MC_INFLAG1: return "INFLAG = 1 OPTION NOT AVAILABLE WITH MCICA";            // src/rrtmg_lw_cldprmc.f90:191
    case E_KISS_PMID: return "MCICA_SUBCOL: KISSVEC SEED GENERATOR REQUIRES PMID FROM BOTTOM FOUR LAYERS.";   // src/mcica_subcol_gen_lw.f90:465
    default: return "unknown physics error";
    }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// the batch size in force for a call of nlay layers
int auto_batch(int nlay)
{
    if (nlay <= 96) return DEFAULT_BATCH;
    int b = DEFAULT_BATCH;
    while (b > 16384 && (long long)b * nlay > (long long)DEFAULT_BATCH * 72) b >>= 1;
    return b;
}
int eff_batch(int nlay)
{
    int b = G.batch_set ? G.batch : auto_batch(nlay);
    // the sweeps address the rows of a workspace array by 32-bit byte offsets (kernels.hip: bload_*'s soff); the largest is that of
    // rtrnmr's overlap factors, (nlay + 1) x 3 rows of 16 bytes per column: a batch that would carry it past 2^32 is halved (262 144
    // columns: beyond 170 layers)
    while (b > 64 && (unsigned long long)(nlay + 2) * 3ull * 16ull * (unsigned long long)b >= (1ull << 32)) b >>= 1;
    return b;
}
// columns per batch: the fewest batches that respect G.batch, of (nearly) equal size - a short last batch would leave the
// pipelines (prep / layer / sweep of neighbouring batches) unbalanced, e.g. 125000 columns -> 2 x 62720 rather than 65536 + 59464
int balanced_batch(int ncol, int cap)
{
    if (ncol <= cap) return ncol;
    const int nbatch = (ncol + cap - 1) / cap;
    const int nb = (int)align_up((size_t)(ncol + nbatch - 1) / nbatch, 256);
    return std::min(nb, cap);
}

hipEvent_t get_event()
{
    if (!G.evpool.empty()) { hipEvent_t e = G.evpool.back(); G.evpool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

// launch wrapper: brackets the kernel with events when profiling is on
#define LAUNCH(NAME, KERNEL, GRID, BLOCKDIM, STREAM, ...)                                        \
    do {                                                                                         \
        if (G.profile) {                                                                         \
            State::ProfRec r_{NAME, get_event(), get_event()};                                   \
            (void)hipEventRecord(r_.a, STREAM);                                                  \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCKDIM, 0, STREAM, __VA_ARGS__);                  \
            (void)hipEventRecord(r_.b, STREAM);                                                  \
            G.prof.push_back(r_);                                                                \
        } else {                                                                                 \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCKDIM, 0, STREAM, __VA_ARGS__);                  \
        }                                                                                        \
    } while (0)

// the same with dynamic LDS
#define LAUNCH_LDS(NAME, KERNEL, GRID, BLOCKDIM, LDS, STREAM, ...)                               \
    do {                                                                                         \
        if (G.profile) {                                                                         \
            State::ProfRec r_{NAME, get_event(), get_event()};                                   \
            (void)hipEventRecord(r_.a, STREAM);                                                  \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCKDIM, LDS, STREAM, __VA_ARGS__);                \
            (void)hipEventRecord(r_.b, STREAM);                                                  \
            G.prof.push_back(r_);                                                                \
        } else {                                                                                 \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCKDIM, LDS, STREAM, __VA_ARGS__);                \
        }                                                                                        \
    } while (0)

// the sweeps stage the transmittance table in LDS: more than the 64 KB a kernel may use without asking
template <int Q>
int sweepc_attr_one()
{
#ifndef RRLW_TUNE
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepc<Q, 0, false, sweepc_nt(Q, 0, false)>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepc<Q, 0, true, sweepc_nt(Q, 0, true)>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepc<Q, 2, true, sweepc_nt(Q, 2, true)>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
#endif
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepc<Q, 1, false, sweepc_nt(Q, 1, false)>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepc<Q, 2, false, sweepc_nt(Q, 2, false)>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    return 0;
}
int ensure_sweep_attrs()
{
    if (G.sweep_attrs) return 0;
#ifdef RRLW_TUNE
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
#else
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<1, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<1, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<1, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<2, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<2, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<3, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<3, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<3, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<3, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<4, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweepz<4, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEPC_LDS_MAX));
#endif
    if (int rc = sweepc_attr_one<1>()) return rc;
    if (int rc = sweepc_attr_one<2>()) return rc;
    if (int rc = sweepc_attr_one<3>()) return rc;
    if (int rc = sweepc_attr_one<4>()) return rc;
    HIP_TRY(hipFuncSetAttribute((const void *)k_flux<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FLUX_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)k_flux<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FLUX_LDS_BYTES));
    G.sweep_attrs = true;
    return 0;
}

// The sweeps' band groups: the bands in [istart, iend] by their number of quads, at most as many per group as one sweep workgroup holds
// (a group is one workgroup in k_sweepc and in k_sweepz: what fits the wave slots of both).  Every group has its own partial slabs.
struct SweepGroups { int n = 0; int nb[NGROUP_MAX]; unsigned long long bands[NGROUP_MAX]; int gq[NGROUP_MAX]; };
bool make_groups(int mode, int idrv, int istart, int iend, SweepGroups &fg)
{
    unsigned long long lists[5] = {0, 0, 0, 0, 0};
    int nbs[5] = {0, 0, 0, 0, 0};
    for (int nq = 1; nq <= 4; nq++)
        for (int B = 1; B <= NBND; B++)
            if (band_nquad(B) == nq && B >= istart && B <= iend) lists[nq] |= (unsigned long long)(B - 1) << (4 * nbs[nq]++);
    fg.n = 0;
    for (int nq = 4; nq >= 1; nq--) {
        const int zcap = sweepz_group_cap(nq, idrv == 1);
        const int cap = std::max(1, std::min(sweepc_group_cap(nq), mode != 0 ? zcap : 99));
        for (int k0 = 0; k0 < nbs[nq]; k0 += cap) {
            if (fg.n >= NGROUP_MAX) return false;
            const int nbg = std::min(cap, nbs[nq] - k0);
            fg.nb[fg.n] = nbg;
            fg.bands[fg.n] = (lists[nq] >> (4 * k0)) & (nbg >= 16 ? ~0ull : ((1ull << (4 * nbg)) - 1ull));
            fg.gq[fg.n++] = nq;
        }
    }
    return true;
}

// (re)allocate the per-batch workspace.  It holds what the call shapes seen so far need and grows with them: the partial slabs of as many
// band groups as the sweeps form (4 without d/dT, up to NGROUP_MAX with), the d/dT slab only for idrv = 1, rtrn's / rtrnmc's emissivity
// term only for modes 1 and 3, rtrnmr's overlap factors only for mode 2.  mode: 0 clear, 1 rtrn, 2 rtrnmr, 3 rtrnmc; -1 = everything.
void graphs_clear();      // (the graphs of small calls hold workspace addresses)
// Batches of up to this many columns are swept ONE band per workgroup (SweepArgs::split; rrtmg_lw_hip_set_split_max / RRTMG_LW_SPLIT_MAX):
// 1 024 columns are 16 blocks of 64 - as groups of bands 16 workgroups of twelve waves on 16 of the 256 CUs, as single bands 256
// workgroups of one to four waves on all of them.  A level of a sweep then costs the dependent issue of ONE wave per SIMD instead of
// three; results bit-identical (k_flux adds a group's bands first).  0 = never.
// Default 768: sixteen bands x twelve blocks = 192 workgroups, each of which wants a CU to itself (the transmittance table takes half of the
// LDS) - at 1 024 columns the 256 workgroups no longer all find one at once, some wait for a second round and the gain is gone; at 4 096
// the rounds make the sweeps 1.6 x slower (profiles/round5_small_calls.md).
int g_split_max = []() { const char *e = getenv("RRTMG_LW_SPLIT_MAX"); return e ? std::max(0, atoi(e)) : 768; }();
int ensure_workspace(int nlay, int ncolb, bool cloud, bool mc = false, int idrv = 1, int mode = -1)
{
    SweepGroups fgw;
    int groups = NGROUP_MAX;
    if (mode >= 0 && make_groups(mode, idrv, 1, NBND, fgw)) groups = fgw.n;
    const bool gdp = idrv == 1, efcl = cloud && (mode < 0 || mode == 1 || mode == 3), ovl = cloud && (mode < 0 || mode == 2);
    // (a batch small enough for the split sweeps - one band per workgroup, a slab per band - needs room for sixteen slabs of its own width)
    const size_t split_cols = ncolb <= g_split_max ? (size_t)NBND * align_up((size_t)ncolb, 64) : 0;
    if (G.ws_base && G.ws_nlay == nlay && G.ws_ncolb >= ncolb && (G.ws_cloud || !cloud) && (G.ws_mc || !mc) && (G.ws_two_scr || !G.split_sweep) &&
        G.ws_groups >= groups && G.ws_slabcols >= split_cols && (G.ws_gdp || !gdp) && (G.ws_efcl || !efcl) && (G.ws_ovl || !ovl)) return 0;
    if (G.ws_base) { HIP_TRY(hipDeviceSynchronize()); graphs_clear(); HIP_TRY(hipFree(G.ws_base)); G.ws_base = nullptr; }
    const bool same = G.ws_nlay == nlay;
    ncolb = std::max(ncolb, same ? G.ws_ncolb : 0);
    cloud = cloud || G.ws_cloud || mc;
    mc = mc || G.ws_mc;
    const int ng = std::max(groups, G.ws_groups);
    const bool want_gdp = gdp || G.ws_gdp, want_efcl = efcl || G.ws_efcl, want_ovl = ovl || G.ws_ovl;
    const size_t n = (size_t)ncolb, L = (size_t)nlay;
    const size_t slabcols = std::max((size_t)ng * n, (size_t)NBND * align_up(std::min(n, (size_t)std::max(g_split_max, 0)), 64));
    struct Item { void **p; size_t bytes; };
    Workspace &W = G.W;
    W = Workspace{};
    std::vector<Item> items = {
        {(void **)&W.gdn1, slabcols * (L + 1) * sizeof(double)},
        {(void **)&W.gup1, slabcols * (L + 1) * sizeof(double)},
        {(void **)&W.gup, slabcols * (L + 1) * sizeof(Part2)},
        {(void **)&W.gdn, slabcols * (L + 1) * sizeof(Part2)},
    };
    if (want_gdp) items.push_back({(void **)&W.gdp, slabcols * (L + 1) * sizeof(Part2)});
    const bool two_scr = G.split_sweep;
    G.scrset[1] = State::ScrSet{};
    for (int k = 0; k < (two_scr ? 2 : 1); k++) {
        State::ScrSet &ss = G.scrset[k];
        ss = State::ScrSet{};
        items.push_back({(void **)&ss.scr[S_CODE], (size_t)NQUAD * L * n * CODE_BYTES});
        items.push_back({(void **)&ss.fw, (size_t)NFW * L * n * sizeof(unsigned)});
        if (cloud) items.push_back({(void **)&ss.scr[S_CODET], (size_t)NQUAD * L * n * CODE_BYTES});
    }
    for (auto &ps : G.prep) {
        ps = State::PrepSet{};
        items.push_back({(void **)&ps.percol, (size_t)NPERCOL * n * 8});
        items.push_back({(void **)&ps.laytrop, n * 4});
        items.push_back({(void **)&ps.ncbands, n * 4});
        items.push_back({(void **)&ps.cflag, (L + 2) * n * 4});
        const size_t nblk = (n + 63) / 64, nslot = (size_t)sort_slots((int)nblk);
        items.push_back({(void **)&ps.btop, nblk * 4});
        items.push_back({(void **)&ps.hblk, nblk * 4});
        items.push_back({(void **)&ps.order, nslot * 4});
        items.push_back({(void **)&ps.hgrp, (nslot / SORT_GROUP) * 4});
        items.push_back({(void **)&ps.bbot, nblk * 4});
        items.push_back({(void **)&ps.hbot, (nslot / SORT_GROUP) * 4});
        items.push_back({(void **)&ps.perm, align_up(n, COLSORT_WIN) * 4});
        items.push_back({(void **)&ps.wsort, (align_up(n, COLSORT_WIN) / COLSORT_WIN) * 4});
        items.push_back({(void **)&ps.wide, (2 + ((n + LAYER_BLOCK - 1) / LAYER_BLOCK) * L) * 4});
        if (cloud) {        // (reordered windows: the temperature and cloud-fraction rows of the caller once more, in position order)
            items.push_back({(void **)&ps.tlayc, L * n * 8});
            items.push_back({(void **)&ps.tlevc, (L + 1) * n * 8});
            items.push_back({(void **)&ps.cldfc, L * n * 8});
        }
    }
    if (cloud) items.push_back({(void **)&W.hand, (size_t)5 * NQUAD * 4 * n * 8});
    if (cloud) {
        for (auto &ps : G.prep) {
            items.push_back({(void **)&ps.odcld, 16 * L * n * 8});
            if (want_efcl) items.push_back({(void **)&ps.efcl, 16 * L * n * 8});
            if (want_ovl) items.push_back({(void **)&ps.ovl, (size_t)2 * (L + 1) * 3 * n * sizeof(double2)});
        }
    }
    if (mc) {
        items.push_back({(void **)&W.odg, (size_t)NQUAD * 4 * L * n * 8});
        items.push_back({(void **)&W.cfef, (size_t)NQUAD * 8 * L * n * 4});
    }
    size_t total = 0;
    for (auto &it : items) total += align_up(it.bytes, 256);
    HIP_TRY(hipMalloc(&G.ws_base, total));
    size_t off = 0;
    for (auto &it : items) { *it.p = (char *)G.ws_base + off; off += align_up(it.bytes, 256); }
    for (auto &ps : G.prep) HIP_TRY(hipMemset(ps.wide, 0, 8));         // the count of k_layer's wide-window list (kernels.hip: k_layer)
    W.ncolb = ncolb;
    W.pcb = ncolb;
    W.nlay = nlay;
    W.err = G.d_err;
    for (int a = 0; a < NSCR; a++) W.scr[a] = G.scrset[0].scr[a];
    W.fw = G.scrset[0].fw;
    {   // G.W itself carries prep set 0 and scratch set 0
        const State::PrepSet &ps = G.prep[0];
        W.percol = ps.percol; W.laytrop = ps.laytrop; W.ncbands = ps.ncbands; W.cflag = ps.cflag;
        W.odcld = ps.odcld; W.efcl = ps.efcl; W.ovl = ps.ovl;
        W.btop = ps.btop; W.order = ps.order; W.hgrp = ps.hgrp; W.hblk = ps.hblk; W.bbot = ps.bbot; W.hbot = ps.hbot;
    }
    G.ws_bytes = total;
    G.ws_nlay = nlay;
    G.ws_ncolb = ncolb;
    G.ws_cloud = cloud;
    G.ws_mc = mc;
    G.ws_two_scr = two_scr;
    G.ws_groups = ng; G.ws_slabcols = slabcols; G.ws_gdp = want_gdp; G.ws_efcl = want_efcl; G.ws_ovl = want_ovl;
    return 0;
}

int ensure_mask(int nlay, size_t ncol)
{
    const size_t bytes = (size_t)KJ_NWORD * nlay * ncol * sizeof(unsigned);
    if (G.mask_bytes < bytes) {
        if (G.mask) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(G.mask)); G.mask = nullptr; G.mask_bytes = 0; }
        HIP_TRY(hipMalloc((void **)&G.mask, bytes));
        G.mask_bytes = bytes;
    }
    return 0;
}

// Columns of a cloudy non-McICA batch are taken in the order k_colsort gives them (by cloud top within windows of 256: the sweeps decide
// per 64 positions where the clouds end).  rrtmg_lw_hip_set_column_sort / RRTMG_LW_COLSORT=0 switch it off; results do not depend on it.
bool g_colsort = []() { const char *e = getenv("RRTMG_LW_COLSORT"); return !e || atoi(e) != 0; }();
bool use_colsort(bool gcm, int mode, int nb);
// block-levels a window's reordering must take out of the cloud zone (k_colsort; measured break-even on an MI355X, profiles/round4_column_order.md)
constexpr int COLSORT_NEVER = 1 << 24;     // no window gains this many block-levels (4 blocks x 603 layers at most): "never reorder"
int g_colsort_min = []() { const char *e = getenv("RRTMG_LW_COLSORT_MIN"); return e ? std::max(0, std::min(atoi(e), COLSORT_NEVER)) : 24; }();

// the workspace view of prep set k (see State::prep); sorted: the batch's columns go through k_colsort's order
Workspace ws_for(int k, bool sorted)
{
    Workspace w = G.W;
    const State::PrepSet &ps = G.prep[k];
    w.perm = sorted ? ps.perm : nullptr;
    w.wsort = sorted ? ps.wsort : nullptr;
    w.tlayc = ps.tlayc; w.tlevc = ps.tlevc; w.cldfc = ps.cldfc;
    w.wide = ps.wide;
    w.percol = ps.percol; w.laytrop = ps.laytrop; w.ncbands = ps.ncbands; w.cflag = ps.cflag;
    w.odcld = ps.odcld; w.efcl = ps.efcl; w.ovl = ps.ovl;
    w.btop = ps.btop; w.order = ps.order; w.hgrp = ps.hgrp; w.hblk = ps.hblk; w.bbot = ps.bbot; w.hbot = ps.hbot;
    const State::ScrSet &ss = G.scrset[G.ws_two_scr ? k : 0];
    for (int a = 0; a < NSCR; a++) w.scr[a] = ss.scr[a];
    w.fw = ss.fw;
    return w;
}

// A stream confined to the CUs [lo, hi) of the device's CU-mask order (hipExtStreamCreateWithCUMask; on a multi-XCD device consecutive
// mask bits go round the XCDs, so a run of bits that is a multiple of 8 long takes the same number of CUs from every XCD), or - lo = hi -
// an ordinary non-blocking stream.
int make_stream(hipStream_t *st, int lo, int hi)
{
    if (hi <= lo) { HIP_TRY(hipStreamCreateWithFlags(st, hipStreamNonBlocking)); return 0; }
    uint32_t mask[32] = {};
    const int nw = (G.cu_total + 31) / 32;
    for (int b = lo; b < hi && b < 1024; b++) mask[b >> 5] |= 1u << (b & 31);
    HIP_TRY(hipExtStreamCreateWithCUMask(st, (uint32_t)nw, mask));
    return 0;
}
int ensure_sweep_set(int k)
{
    State::SweepSet &SS = G.swset[k];
    if (SS.q[0]) return 0;
    const int lo = k == 1 ? G.cu_layer : 0, hi = k == 1 ? G.cu_total : 0;
    if (!SS.main) { if (int rc = make_stream(&SS.main, lo, hi)) return rc; }
    for (int j = 0; j < 3; j++) {
        if (int rc = make_stream(&SS.q[j], lo, hi)) return rc;
        HIP_TRY(hipEventCreateWithFlags(&SS.done[j], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&SS.go, hipEventDisableTiming));
    return 0;
}
void drop_sweep_set(int k)
{
    State::SweepSet &SS = G.swset[k];
    if (SS.main) (void)hipStreamDestroy(SS.main);
    for (int j = 0; j < 3; j++) { if (SS.q[j]) (void)hipStreamDestroy(SS.q[j]); if (SS.done[j]) (void)hipEventDestroy(SS.done[j]); }
    if (SS.go) (void)hipEventDestroy(SS.go);
    SS = State::SweepSet{};
}

// A cloudy batch too small to fill the chip is a latency chain of six kernels; three of them are the sweeps (down above the clouds, the
// cloud zone, up above the clouds).  Up to ONE_SWEEP_MAX columns the cloud-zone kernel walks all levels instead (k_blocksort, force_top)
// and the two clear-sky launches are not made: 1 024 cloudy columns 0.609 -> 0.587 ms (profiles/round4_small_calls.md).  Larger batches keep the three launches - above the
// clouds k_sweepc costs a third of k_sweepz's clear-sky body per level.
int g_one_sweep_max = []() { const char *e = getenv("RRTMG_LW_ONE_SWEEP_MAX"); return e ? atoi(e) : 4096; }();       // rrtmg_lw_hip_set_one_sweep_max
bool one_sweep(int nb, int mode) { return mode != 0 && nb <= g_one_sweep_max; }
// k_layer's second pass with the wide staging window (terrain-following pressure grids; kernels.hip: StageWin).  rrtmg_lw_hip_set_wide_window /
// RRTMG_LW_WIDE_WINDOW=0 switch it off (measurement: every workgroup then keeps the narrow window, as before round 5); same results either way.
bool g_wide_window = []() { const char *e = getenv("RRTMG_LW_WIDE_WINDOW"); return !e || atoi(e) != 0; }();
// k_layer's bands of a (window, layer) over several workgroups where the batch does not fill the chip (run_layer).  rrtmg_lw_hip_set_layer_split /
// RRTMG_LW_LAYER_SPLIT=0.  Results do not depend on it.
bool g_prep_fork = []() { const char *e = getenv("RRTMG_LW_PREP_FORK"); return !e || atoi(e) != 0; }();     // (run_prep: k_colprep beside k_cloudscan; measurement switch)
bool g_layer_split = []() { const char *e = getenv("RRTMG_LW_LAYER_SPLIT"); return !e || atoi(e) != 0; }();
// (a batch that takes one sweep launch walks every level in the cloud-zone kernel whatever its blocks hold: nothing to gain from an order)
// (McICA, mode 3: with the generator's mask - the grid-mean cloud fraction gives the key; the sub-column ARRAYS of the reference's McICA
// argument list come without one and keep their order: the call sites pass `!mc`)
bool use_colsort(bool gcm, int mode, int nb) { return g_colsort && gcm && (mode == 1 || mode == 2 || mode == 3) && !one_sweep(nb, mode); }

// per-column part of one batch: k_colprep (+ k_cloudscan / k_cloudlay for rtrn / rtrnmr): it runs on the
// auxiliary stream one batch ahead of the heavy kernels (run_pipelined).
template <bool GCM>
int run_prep(hipStream_t s, const Workspace &Wk, int nb, int col0, int nct, int mode, int idrv, int istart,
             const GcmIn &g, const ColIn &c, int inflag, int iceflag, int liqflag, hipStream_t side = nullptr)
{
    // thread-per-column kernels: one wave per workgroup so that a batch (one wave per 64 columns) spreads over all 256 CUs
    const dim3 cgrid1((nb + 63) / 64), cblock1(64);
    if (Wk.perm) {
        const dim3 wgrid((nb + COLSORT_WIN - 1) / COLSORT_WIN);
        LAUNCH("k_colsort", (k_colsort<GCM>), wgrid, dim3(COLSORT_WIN, COLSORT_TY), s, Wk, g, c, nb, col0, nct, g_colsort_min);
    }
    // (k_cloudscan needs nothing of k_colprep - k_cloudlay does: the secants - and both are one dependent walk over a column's layers: with a
    // side stream, a call that is a single batch, they run side by side: -30 us of its 0.5 ms)
    const bool scan = mode == 1 || mode == 2, fork = scan && side && side != s;
    if (fork) {
        HIP_TRY(hipEventRecord(G.ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(side, G.ev_fork, 0));
    }
    LAUNCH("k_colprep", (k_colprep<GCM>), cgrid1, cblock1, fork ? side : s, G.D, Wk, g, c, nb, col0, nct, idrv, istart, scan ? 0 : 1);
    if (fork) HIP_TRY(hipEventRecord(G.ev_join, side));
    if (scan) {
        LAUNCH("k_cloudscan", (k_cloudscan<GCM>), cgrid1, cblock1, s, Wk, g, c, nb, col0, nct, inflag, iceflag, liqflag, mode);
        if (fork) HIP_TRY(hipStreamWaitEvent(s, G.ev_join, 0));
        const dim3 lgrid((nb + BLOCK - 1) / BLOCK, Wk.nlay), lblock(BLOCK);
        LAUNCH("k_cloudlay", (k_cloudlay<GCM>), lgrid, lblock, s, G.D, Wk, g, c, nb, col0, nct, mode, inflag, iceflag, liqflag);
        LAUNCH("k_blocksort", k_blocksort, dim3(1), dim3(256), s, Wk, (nb + 63) / 64, one_sweep(nb, mode) ? 1 : 0);      // the blocks by cloud top, hand-off levels
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RRTMG_LW_HIP_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

// layer-local part of one batch (k_cloudmc, k_layer groups), everything device-resident.  mode: 0 clear, 1 rtrn, 2 rtrnmr,
// 3 rtrnmc (McICA; `mc` = the sub-column arrays, or null when the sub-columns come from the generator's mask in Wk.mask)
template <bool GCM>
int run_layer(hipStream_t s, const Workspace &Wk, int nb, int col0, int nct, int nlay, int mode, int idrv, int istart, int iend,
              const GcmIn &g, const ColIn &c, int inflag, int iceflag, int liqflag, const McIn *mc = nullptr)
{
    const dim3 block(BLOCK);
    if (mode == 3) {
        const dim3 cgrid((nb + BLOCK - 1) / BLOCK, nlay);
        if (mc) LAUNCH("k_cloudmc<arrays>", (k_cloudmc<false>), cgrid, block, s, G.D, Wk, *mc, g, nb, col0, nct, inflag, iceflag, liqflag);
        else LAUNCH("k_cloudmc<mask>", (k_cloudmc<true>), cgrid, block, s, G.D, Wk, McIn{}, g, nb, col0, nct, inflag, iceflag, liqflag);
        LAUNCH("k_blocksort", k_blocksort, dim3(1), dim3(256), s, Wk, (nb + 63) / 64, one_sweep(nb, mode) ? 1 : 0);
    }
    LayerArgs la;
    la.ncol = nb; la.col0 = col0; la.nct = nct; la.idrv = idrv; la.istart = istart; la.iend = iend;
    la.ktab_bytes = (int)(G.H.ktab.size() * 8);
    la.tauaer = GCM ? g.tauaer : c.taua;
    const unsigned gx = (nb + LAYER_BLOCK - 1) / LAYER_BLOCK;
    // A batch whose (window, layer) pairs do not fill the chip - 768 workgroups at three per CU - spreads the bands of a pair over two or
    // four workgroups (k_layer: LayerArgs::partmask): a workgroup's sixteen bands are a 100 us chain, and a small call waits for ONE round
    // of them.  The parts follow the staging passes (a workgroup stages only the passes it has a band of).
    const auto bits = [](std::initializer_list<int> bands) { unsigned m = 0; for (int b : bands) m |= 1u << (b - 1); return m; };
    const unsigned quarter[4] = {bits({1, 2, 11, 15, 6}), bits({8, 10, 14, 16, 12, 13}), bits({4, 9, 7}), bits({3, 5})};
    const unsigned pairs_ = gx * (unsigned)nlay;
    la.nparts = !g_layer_split ? 1 : (pairs_ * 4 <= 1152 ? 4 : (pairs_ * 2 <= 1152 ? 2 : 1));
    for (int p = 0; p < 4; p++) la.partmask[p] = la.nparts == 4 ? quarter[p] : (la.nparts == 2 && p < 2 ? (quarter[2 * p] | quarter[2 * p + 1]) : 0xffffu);
    const dim3 lgrid(gx, nlay, la.nparts), lblock(LAYER_BLOCK);
    // the wide-window pass over the (window, layer) pairs the narrow launch could not take (GCM entry; kernels.hip: StageWin): as many
    // workgroups as fit the chip at once, which leave at once when the list is empty
    Workspace Wn = Wk;
    if (!GCM || !HAVE_WIDE || !g_wide_window) Wn.wide = nullptr;
    const dim3 wgrid(gx * nlay, la.nparts);
    // (the list's count: cleared on the stream in front of the narrow launch - a memset node when the call is captured as a graph)
    if (Wn.wide) HIP_TRY(hipMemsetAsync(Wn.wide, 0, sizeof(int), s));
#ifdef RRLW_TUNE
    // tuning builds (tools/build_variant.sh name -DRRLW_TUNE ...): only the kernels of the benchmark's default workload - GCM entry,
    // rtrn / rtrnmr, idrv = 0 - are instantiated (a quarter of the compile time); every other call shape is refused
#define LAYER_GROUP(GR)                                                                                              \
    if constexpr (GCM) { if (mode == 1 || mode == 2) { LAUNCH("k_layer<cloud," #GR ">", (k_layer<true, 1, GR>), lgrid, lblock, s, G.D, Wn, g, c, la); \
                                                       if (Wn.wide) LAUNCH("k_layer<cloud,1>", (k_layer<true, 1, 1>), wgrid, lblock, s, G.D, Wn, g, c, la); } \
                         else return fail(RRTMG_LW_HIP_EARG, "tuning build: cloudy non-McICA calls only"); }          \
    else return fail(RRTMG_LW_HIP_EARG, "tuning build: GCM entry only");
#else
#define LAYER_GROUP(GR)                                                                                              \
    if (mode == 0) LAUNCH("k_layer<clear," #GR ">", (k_layer<GCM, 0, GR>), lgrid, lblock, s, G.D, Wn, g, c, la);         \
    else if (mode == 3) { if (mc) LAUNCH("k_layer<mcica," #GR ">", (k_layer<GCM, 2, GR>), lgrid, lblock, s, G.D, Wn, g, c, la); \
                          else if constexpr (GCM) LAUNCH("k_layer<mcmask," #GR ">", (k_layer<true, 3, GR>), lgrid, lblock, s, G.D, Wn, g, c, la); } \
    else LAUNCH("k_layer<cloud," #GR ">", (k_layer<GCM, 1, GR>), lgrid, lblock, s, G.D, Wn, g, c, la);             \
    if constexpr (GCM) {                                                                                             \
        if (Wn.wide) {                                                                                               \
            if (mode == 0) LAUNCH("k_layer<clear,1>", (k_layer<true, 0, 1>), wgrid, lblock, s, G.D, Wn, g, c, la);        \
            else if (mode == 3) { if (mc) LAUNCH("k_layer<mcica,1>", (k_layer<true, 2, 1>), wgrid, lblock, s, G.D, Wn, g, c, la); \
                                  else LAUNCH("k_layer<mcmask,1>", (k_layer<true, 3, 1>), wgrid, lblock, s, G.D, Wn, g, c, la); } \
            else LAUNCH("k_layer<cloud,1>", (k_layer<true, 1, 1>), wgrid, lblock, s, G.D, Wn, g, c, la);                  \
        }                                                                                                            \
    }
#endif
    LAYER_GROUP(0)
#undef LAYER_GROUP
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RRTMG_LW_HIP_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

// vertical part of one batch: the sweep launches and k_flux
template <bool GCM>
int run_sweep(hipStream_t s, const Workspace &Wk_, int nb, int col0, int nct, int nlay, int mode, int idrv, int istart, int iend,
              const GcmIn &g, const ColIn &c, const FluxOut &out, const McIn *mc = nullptr, int sset = 0)
{
    Workspace Wk = Wk_;
    // a batch that does not fill the chip: one band per workgroup, a slab per band (g_split_max, SweepArgs::split) - where the slab arrays hold
    // sixteen slabs of the batch's width
    const size_t pcb_split = align_up((size_t)nb, 64);
    const bool split = nb <= g_split_max && (size_t)NBND * pcb_split <= G.ws_slabcols && pcb_split <= (size_t)Wk.ncolb;
    Wk.pcb = split ? (int)pcb_split : Wk.ncolb;
    const dim3 block(BLOCK);
    if (int rc = ensure_sweep_attrs()) return rc;
    // The four sweep launches of a batch (bands of 4, 3, 2, 1 quads) are independent.  Each workgroup owns a CU, so a launch ends with
    // a partly filled last round (1-quad bands: 2.5 rounds); on separate streams the other launches' workgroups fill those CUs.
    // Measured: 10 000 columns 1.22 -> 0.98 ms; 125 000-column batches no gain (cloudy 98.9 -> 98.4 ms per 1e6 columns, McICA 115.5 -> 122.6).
#ifndef RRLW_FANOUT_MAX
#define RRLW_FANOUT_MAX 0x7fffffff     // (round 3: for every batch size - 61.4 -> 60.2 ms per 1e6 cloudy columns, three runs each; deep clouds 87.7 -> 85.7.
#endif                                 // The HIP-event times of the sweep kernels then overlap: their sum exceeds the wall time they take together)
    const bool fan = G.sweep_fanout && nb < RRLW_FANOUT_MAX;
    State::SweepSet &SS = G.swset[sset];
    if (fan) {
        if (int rc = ensure_sweep_set(sset)) return rc;
        HIP_TRY(hipEventRecord(SS.go, s));
        for (int k = 0; k < 3; k++) HIP_TRY(hipStreamWaitEvent(SS.q[k], SS.go, 0));
    }
    const hipStream_t s_main = s;
#ifndef RRLW_TUNE
    if (G.n1 && mode == 0 && GCM && istart == 1 && iend == 16 && idrv == 0 && n1_lds_bytes(nlay) <= 160 * 1024) {
        // prototype of the north-star mapping (one column per wavefront): replaces the sweeps, k_flux and k_rates of a cloud-free call
        SweepArgs sa{};
        sa.ncol = nb; sa.col0 = col0; sa.nct = nct; sa.idrv = 0; sa.istart = 1; sa.iend = 16;
        sa.emis = g.emis; sa.cldfrac = nullptr; sa.tlay = g.tlay; sa.tlev = g.tlev;
        static bool attr = false;
        if (!attr) { HIP_TRY(hipFuncSetAttribute((const void *)k_n1<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
        const dim3 ngrid((nb + N1_WAVES - 1) / N1_WAVES), nblock(64 * N1_WAVES);
        LAUNCH_LDS("k_n1", (k_n1<false>), ngrid, nblock, n1_lds_bytes(nlay), s, G.D, Wk, sa, out, g.plev);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(RRTMG_LW_HIP_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
        return 0;
    }
#endif
    SweepArgs sa;
    sa.ncol = nb; sa.col0 = col0; sa.nct = nct; sa.idrv = idrv; sa.istart = istart; sa.iend = iend;
    sa.split = split ? 1 : 0;
    sa.emis = GCM ? g.emis : c.semiss;
    sa.cldfrac = GCM ? g.cldfr : c.cldfrac;
    sa.tlay = GCM ? g.tlay : c.tavel;
    sa.tlev = GCM ? g.tlev : c.tz;
    // Per class of bands with the same number of quads: a cloud-free call (mode 0) is one k_sweepc<., 0> launch; the cloudy modes run
    // k_sweepc<., 1> (layers above the batch's highest cloud, downward), k_sweepz<., mode> (layers 1 .. ltop down, surface, up) and
    // k_sweepc<., 2> (layers above, upward).  The classes are independent of each other: with `fan` each class has its own stream;
    // on one stream the launches go phase by phase (all downward ones, then the cloud zone, then the upward ones) so that consecutive
    // launches never wait for each other's last workgroups.
    SweepGroups fg;
    if (!make_groups(mode, idrv, istart, iend, fg)) return fail(RRTMG_LW_HIP_EARG, "internal: more than %d sweep groups", NGROUP_MAX);
    if ((!split && fg.n > G.ws_groups) || (idrv == 1 && !G.ws_gdp)) return fail(RRTMG_LW_HIP_EARG, "internal: workspace holds %d band groups, the call needs %d", G.ws_groups, fg.n);
    const int *gq = fg.gq;
    // first slab of every group, slabs per group (k_flux)
    int slab0[NGROUP_MAX];
    unsigned long long gsz = 0ull;
    { int sl = 0; for (int g = 0; g < fg.n; g++) { slab0[g] = sl; const int n = split ? fg.nb[g] : 1; gsz |= (unsigned long long)n << (4 * g); sl += n; } }
#define SWEEPC_I(Q, PH, I)                                                                                           \
    do {                                                                                                             \
        constexpr int nt = sweepc_nt(Q, PH, I);                                                                      \
        const int wnb = split ? 1 : sa.nbands;              /* bands of a workgroup */                               \
        const int nsb = split ? 1 : sweepc_nsb(Q, PH, I, sa.nbands);                                                 \
        sa.ncb = (nb + 64 * nsb - 1) / (64 * nsb);                                                                   \
        const dim3 sgrid((unsigned)sa.ncb, split ? sa.nbands : 1), sblock(64, wnb * nt, nsb);                        \
        LAUNCH_LDS("k_sweepc<" #Q "," #PH ">", (k_sweepc<Q, PH, I, nt>), sgrid, sblock, sweepc_lds_bytes(PH, I, wnb, nsb, nt), s, G.D, Wk, sa); \
    } while (0)
#ifdef RRLW_TUNE
#define SWEEPC(Q, PH) do { if (idrv == 1) return fail(RRTMG_LW_HIP_EARG, "tuning build: idrv = 0 only"); else SWEEPC_I(Q, PH, false); } while (0)
#else
#define SWEEPC(Q, PH) do { if (idrv == 1 && PH != 1) SWEEPC_I(Q, PH, true); else SWEEPC_I(Q, PH, false); } while (0)
#endif
#define SWEEPC_Q(PH) do { if (nq == 4) SWEEPC(4, PH); else if (nq == 3) SWEEPC(3, PH); else if (nq == 2) SWEEPC(2, PH); else SWEEPC(1, PH); } while (0)
    for (int phase = 0; phase < 3; phase++) {
        if (phase == 1) {                           // cloud zone: k_sweepz, one launch per group
            if (mode == 0) continue;
            for (int g = 0; g < fg.n; g++) {
                const int nq = gq[g];
                sa.bands = fg.bands[g];
                sa.nbands = fg.nb[g];
                sa.group = slab0[g];
                const hipStream_t s = (fan && nq < 4) ? SS.q[3 - nq] : s_main;
#define SWEEPZ_I(Q, M, I)                                                                                            \
    do {                                                                                                             \
        constexpr int nt = sweepz_nt(Q);                                                                             \
        const int wnb = split ? 1 : sa.nbands;                                                                       \
        const int nsb = split ? 1 : sweepz_nsb(Q, sa.nbands, I);                                                     \
        sa.ncb = (nb + 64 * nsb - 1) / (64 * nsb);                                                                   \
        const dim3 sgrid((unsigned)sa.ncb, split ? sa.nbands : 1), sblock(64, wnb * nt, nsb);                        \
        LAUNCH_LDS("k_sweepz<" #Q "," #M ">", (k_sweepz<Q, M, I>), sgrid, sblock, sweepz_lds_bytes(wnb, nsb, nt, I), s, G.D, Wk, sa); \
    } while (0)
#ifdef RRLW_TUNE
#define SWEEPZ_M(Q, M) do { if (idrv == 1) return fail(RRTMG_LW_HIP_EARG, "tuning build: idrv = 0 only"); else SWEEPZ_I(Q, M, false); } while (0)
#define SWEEPZ(Q) do { if (mode == 2) SWEEPZ_M(Q, 2); else return fail(RRTMG_LW_HIP_EARG, "tuning build: rtrnmr only"); } while (0)
#else
#define SWEEPZ_M(Q, M) do { if (idrv == 1) SWEEPZ_I(Q, M, true); else SWEEPZ_I(Q, M, false); } while (0)
#define SWEEPZ(Q) do { if (mode == 1) SWEEPZ_M(Q, 1); else if (mode == 3 && mc) SWEEPZ_M(Q, 3); else if (mode == 3) SWEEPZ_M(Q, 4); else SWEEPZ_M(Q, 2); } while (0)
#endif
                if (nq == 4) SWEEPZ(4); else if (nq == 3) SWEEPZ(3); else if (nq == 2) SWEEPZ(2); else SWEEPZ(1);
#undef SWEEPZ_I
#undef SWEEPZ_M
#undef SWEEPZ
            }
            continue;
        }
        if (mode == 0 && phase == 2) continue;
        if (one_sweep(nb, mode)) continue;          // (the cloud-zone kernel walks all levels of a small batch)
        for (int g = 0; g < fg.n; g++) {            // above the clouds (or a cloud-free call): one launch per group
            const int nq = gq[g];
            sa.bands = fg.bands[g];
            sa.nbands = fg.nb[g];
            sa.group = slab0[g];
            const hipStream_t s = (fan && nq < 4) ? SS.q[3 - nq] : s_main;
#ifdef RRLW_TUNE
            if (mode == 0) return fail(RRTMG_LW_HIP_EARG, "tuning build: cloudy calls only");
#else
            if (mode == 0) SWEEPC_Q(0);
#endif
            else if (phase == 0) SWEEPC_Q(1);
            else SWEEPC_Q(2);
        }
    }
#undef SWEEPC_Q
#undef SWEEPC
#undef SWEEPC_I
    if (fan) {
        for (int k = 0; k < 3; k++) {
            HIP_TRY(hipEventRecord(SS.done[k], SS.q[k]));
            HIP_TRY(hipStreamWaitEvent(s_main, SS.done[k], 0));
        }
    }
    {
        const dim3 wgrid((nb + COLSORT_WIN - 1) / COLSORT_WIN, (nlay + 1 + FLUX_LV - 1) / FLUX_LV), wblock(COLSORT_WIN, FLUX_TY);
        const double *pz = GCM ? g.plev : c.pz;
        if (idrv == 1) LAUNCH_LDS("k_flux", (k_flux<true>), wgrid, wblock, FLUX_LDS_BYTES, s, G.D, Wk, out, pz, nb, col0, nct, mode == 0 ? 1 : 0, fg.n, gsz);
        else LAUNCH_LDS("k_flux", (k_flux<false>), wgrid, wblock, FLUX_LDS_BYTES, s, G.D, Wk, out, pz, nb, col0, nct, mode == 0 ? 1 : 0, fg.n, gsz);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RRTMG_LW_HIP_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}


// one batch on one stream (host-pointer entries, whose staging buffer serialises the batches anyway)
template <bool GCM>
int run_batch(hipStream_t s, int nb, int col0, int nct, int nlay, int mode, int idrv, int istart, int iend,
              const GcmIn &g, const ColIn &c, int inflag, int iceflag, int liqflag, const FluxOut &out, const McIn *mc = nullptr)
{
    const Workspace Wk = ws_for(0, use_colsort(GCM, mode, nb) && !mc);
    if (int rc = run_prep<GCM>(s, Wk, nb, col0, nct, mode, idrv, istart, g, c, inflag, iceflag, liqflag)) return rc;
    if (int rc = run_layer<GCM>(s, Wk, nb, col0, nct, nlay, mode, idrv, istart, iend, g, c, inflag, iceflag, liqflag, mc)) return rc;
    return run_sweep<GCM>(s, Wk, nb, col0, nct, nlay, mode, idrv, istart, iend, g, c, out, mc);
}

int ensure_pipeline()
{
    if (G.aux) return 0;
    HIP_TRY(hipStreamCreateWithFlags(&G.aux, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&G.ev_in, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&G.ev_last, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&G.ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&G.ev_join, hipEventDisableTiming));
    for (int k = 0; k < 2; k++) {
        HIP_TRY(hipEventCreateWithFlags(&G.ev_ready[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&G.ev_layer[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&G.ev_done[k], hipEventDisableTiming));
    }
    return 0;
}

// all batches of a device-resident call on the caller's stream `s`, the per-column kernels of batch i+1 overlapping the
// heavy kernels of batch i on the auxiliary stream (two prep sets)
// Build switch: hold the sub-column generator of batch i+1 back until k_layer of batch i is done, so that it runs beside the sweeps
// instead of beside k_layer.  Measured per 1e6 McICA columns: beside k_layer the generator takes 19.3 ms and k_layer 35.7 (26.7 alone,
// both VALU-bound), beside the sweeps the generator takes its 12.9 ms and k_sweepc 29.1 instead of 19.9 (a sweep work-group needs most
// of a CU's LDS and waits for the generator's work-groups to leave) - 77.5 vs 77.7 ms a step either way, also with the auxiliary stream
// at the lowest priority.  The generator's 12.9 ms are added work, not hidden work.
#ifndef RRLW_GEN_BESIDE_SWEEP
#define RRLW_GEN_BESIDE_SWEEP 0
#endif
int launch_kiss(hipStream_t s, const Workspace &Wk, int ncol, int col0, int nb, int nlay, int icld, int permuteseed, const SubcolIn &in);

// A device-resident call of ONE batch that does not fill the chip is a chain of a dozen dependent launches on up to four streams: the
// kernels themselves take two thirds of its time, launch gaps and cross-stream event hops the rest (profiles/round4_small_calls.md).  The
// second call with the same arguments - shape, flags, every array pointer: a host model hands over the same arrays time step after time
// step - is captured (hipStreamBeginCapture on a stream of the library's own) and every later one replays the instantiated graph with one
// hipGraphLaunch on the caller's stream.  rrtmg_lw_hip_set_graph_max / RRTMG_LW_GRAPH_MAX: the largest call (columns) taken this way;
// 0 = never.  Same kernels, same arguments: same results.
int g_graph_max = []() { const char *e = getenv("RRTMG_LW_GRAPH_MAX"); return e ? std::max(0, atoi(e)) : 16384; }();
constexpr size_t GRAPH_CACHE = 8;
void graphs_clear()
{
    for (auto &e : G.graphs) if (e.exec) (void)hipGraphExecDestroy(e.exec);
    G.graphs.clear();
}
template <class T> void key_put(std::vector<unsigned char> &k, const T &v)
{
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&v);
    k.insert(k.end(), p, p + sizeof(T));
}

struct KissGen { bool on; int icld, permuteseed; const double *alpha; };      // kissvec generator folded into the per-batch prep

int run_pipelined(hipStream_t s, int ncol, int nlay, int mode, int idrv, const GcmIn &g, int inflag, int iceflag, int liqflag,
                  const FluxOut &out, const McIn *mc, KissGen gen = KissGen{false, 0, 0, nullptr})
{
    if (int rc = ensure_pipeline()) return rc;
    const int nbmax = balanced_batch(ncol, eff_batch(nlay));
    // a call that is ONE batch has nothing for its per-column kernels to run beside: they go on the caller's stream, one cross-stream
    // event hop (~20 us of a 0.6 ms call) less
    const bool single = ncol <= nbmax && !(G.split_sweep && G.cu_layer > 0);
    const hipStream_t aux = single ? s : G.aux;
    ColIn c{};
    // sub-column arrays mode keeps per-g-point cloud arrays (odg/cfef) in one set only: no layer/sweep overlap there
    const bool split = G.split_sweep && G.ws_two_scr && !(mode == 3 && mc);
    const bool part = split && G.cu_layer > 0;
    if (part && !G.lay_m) {
        if (int rc = make_stream(&G.lay_u, 0, 0)) return rc;
        if (int rc = make_stream(&G.lay_m, 0, G.cu_layer)) return rc;
    }
    if (G.ev_last_valid) HIP_TRY(hipStreamWaitEvent(s, G.ev_last, 0));      // an earlier call, possibly on another stream, still owns the workspace
    // a small one-batch call as a graph (above): replay, capture (second call with this key), or the plain launches
    State::GraphEnt *gent = nullptr;
    bool capture = false;
    if (single && !split && !gen.on && !mc && !G.profile && !G.n1 && ncol <= g_graph_max) {
        std::vector<unsigned char> key;
        key_put(key, ncol); key_put(key, nlay); key_put(key, mode); key_put(key, idrv); key_put(key, inflag); key_put(key, iceflag); key_put(key, liqflag);
        const double *gp[] = {g.play, g.plev, g.tlay, g.tlev, g.tsfc, g.h2ovmr, g.o3vmr, g.co2vmr, g.ch4vmr, g.n2ovmr, g.o2vmr, g.cfc11vmr, g.cfc12vmr,
                              g.cfc22vmr, g.ccl4vmr, g.emis, g.cldfr, g.taucld, g.cicewp, g.cliqwp, g.reice, g.reliq, g.tauaer, g.tauctot,
                              out.uflx, out.dflx, out.hr, out.uflxc, out.dflxc, out.hrc, out.duflx_dt, out.duflxc_dt, out.fnet, out.fnetc};
        key_put(key, gp);
        // (what else decides which kernels run with which arguments: the workspace, the tuning switches)
        key_put(key, G.ws_base); key_put(key, G.ws_bytes); key_put(key, g_colsort); key_put(key, g_colsort_min); key_put(key, g_one_sweep_max);
        key_put(key, g_wide_window); key_put(key, g_layer_split); key_put(key, g_prep_fork); key_put(key, G.sweep_fanout); key_put(key, eff_batch(nlay)); key_put(key, g_split_max);
        for (auto &e : G.graphs) if (e.key == key) { gent = &e; break; }
        if (!gent) {
            if (G.graphs.size() >= GRAPH_CACHE) {           // the entry used longest ago makes room
                size_t old = 0;
                for (size_t j = 1; j < G.graphs.size(); j++) if (G.graphs[j].used < G.graphs[old].used) old = j;
                if (G.graphs[old].exec) (void)hipGraphExecDestroy(G.graphs[old].exec);
                G.graphs.erase(G.graphs.begin() + (long)old);
            }
            G.graphs.emplace_back();
            gent = &G.graphs.back();
            gent->key = key;
        }
        gent->used = ++G.graph_clock;
        if (gent->exec) {
            HIP_TRY(hipGraphLaunch(gent->exec, s));
            G.graph_replays++;
            HIP_TRY(hipEventRecord(G.ev_last, s));
            G.ev_last_valid = true;
            return 0;
        }
        // the first call with a key takes the plain launches (whatever is set up lazily - streams, kernel attributes - then exists), the
        // second is captured
        capture = gent->seen >= 1 && !gent->failed;
        gent->seen++;
        if (capture && !G.cap) HIP_TRY(hipStreamCreateWithFlags(&G.cap, hipStreamNonBlocking));
    }
    const hipStream_t s_call = s;
    if (capture) {
        if (hipStreamBeginCapture(G.cap, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); capture = false; gent->failed = true; }
        else s = G.cap;
    }
    const hipStream_t aux_ = capture ? s : aux;
    if (!single) {
        HIP_TRY(hipEventRecord(G.ev_in, s));             // inputs are ready when the caller's stream gets here
        HIP_TRY(hipStreamWaitEvent(aux, G.ev_in, 0));
    }
    int i = 0, rc_cap = 0;
    for (int col0 = 0; col0 < ncol; col0 += nbmax, i++) {
        const int nb = std::min(nbmax, ncol - col0), k = i & 1;
        const Workspace Wk = ws_for(k, use_colsort(true, mode, nb) && !mc);
        if (i >= 2) HIP_TRY(hipStreamWaitEvent(aux, G.ev_done[k], 0));       // prep set k is free again (sweep of batch i-2 done)
        // (The per-column kernels of batch i thus run beside k_layer of batch i-1.  Their few long-lived waves cost whatever runs beside
        // them about what the overlap saves - measured per 1e6 columns: k_layer 32.7 ms beside them, 28.3 alone, step 93.9 vs 96.0 on one
        // stream; held back until k_layer is done they slow the sweep instead, 94.4, and the McICA generator then costs 7 ms more.)
        if (capture) {
            // (one batch, one stream: the batch's launches, the fork to the sweep streams and their join become the graph; an error ends
            // the capture before it is reported)
            rc_cap = run_prep<true>(s, Wk, nb, col0, ncol, mode, idrv, 1, g, c, inflag, iceflag, liqflag, g_prep_fork ? G.aux : nullptr);
            if (rc_cap == 0) rc_cap = run_layer<true>(s, Wk, nb, col0, ncol, nlay, mode, idrv, 1, 16, g, c, inflag, iceflag, liqflag, mc);
            if (rc_cap == 0) rc_cap = run_sweep<true>(s, Wk, nb, col0, ncol, nlay, mode, idrv, 1, 16, g, c, out, mc);
            continue;
        }
        if (int rc = run_prep<true>(aux, Wk, nb, col0, ncol, mode, idrv, 1, g, c, inflag, iceflag, liqflag, single && g_prep_fork ? G.aux : nullptr)) return rc;
        if (gen.on) {
#if RRLW_GEN_BESIDE_SWEEP
            if (i >= 1 && !split) HIP_TRY(hipStreamWaitEvent(aux, G.ev_layer[(i - 1) & 1], 0));
#endif
            if (int rc = launch_kiss(aux, Wk, ncol, col0, nb, nlay, gen.icld, gen.permuteseed, SubcolIn{g.play, g.cldfr, gen.alpha})) return rc;
        }
        if (!single) HIP_TRY(hipEventRecord(G.ev_ready[k], aux));
        // (with a CU partition nothing is enqueued on the caller's stream inside the loop: streams with a CU mask are blocking streams, and
        // where the caller's stream is the null stream every operation on it would be a barrier between them)
        if (!part && !single) HIP_TRY(hipStreamWaitEvent(s, G.ev_ready[k], 0));
        if (split) {
            // k_layer of this batch on the caller's stream, its sweep on the sweep set's main stream: the HBM-bound sweep of batch i
            // overlaps the issue-bound k_layer of batch i+1 (scratch set k is free once the sweep of batch i-2 is done).
            // With a CU partition (rrtmg_lw_hip_set_cu_partition) the two run on their own CUs: k_layer's workgroups otherwise take every
            // register and LDS slot of whichever CU they reach first and the sweep workgroups (one per CU: the transmittance table) trickle
            // in behind them.  The first batch's k_layer and the last batch's sweeps have nothing beside them and take the whole chip.
            const bool last = col0 + nbmax >= ncol;
            hipStream_t sl = s;
            int sset = 0;
            if (part) { sl = i == 0 ? G.lay_u : G.lay_m; sset = last ? 0 : 1; }
            if (int rc = ensure_sweep_set(sset)) return rc;
            const hipStream_t ssm = G.swset[sset].main;
            if (part) {
                if (i == 0) HIP_TRY(hipStreamWaitEvent(sl, G.ev_in, 0));
                HIP_TRY(hipStreamWaitEvent(sl, G.ev_ready[k], 0));
                if (i >= 1) HIP_TRY(hipStreamWaitEvent(sl, G.ev_layer[(i - 1) & 1], 0));     // (k_layer launches stay in batch order across the two streams)
            }
            if (i >= 2) HIP_TRY(hipStreamWaitEvent(sl, G.ev_done[k], 0));
            if (int rc = run_layer<true>(sl, Wk, nb, col0, ncol, nlay, mode, idrv, 1, 16, g, c, inflag, iceflag, liqflag, mc)) return rc;
            HIP_TRY(hipEventRecord(G.ev_layer[k], sl));
            HIP_TRY(hipStreamWaitEvent(ssm, G.ev_layer[k], 0));
            if (part && i >= 1) HIP_TRY(hipStreamWaitEvent(ssm, G.ev_done[(i - 1) & 1], 0));   // (the partial slabs and the hand-off array exist once: sweeps stay in batch order across the two sets)
            if (int rc = run_sweep<true>(ssm, Wk, nb, col0, ncol, nlay, mode, idrv, 1, 16, g, c, out, mc, sset)) return rc;
            HIP_TRY(hipEventRecord(G.ev_done[k], ssm));
        } else {
            if (int rc = run_layer<true>(s, Wk, nb, col0, ncol, nlay, mode, idrv, 1, 16, g, c, inflag, iceflag, liqflag, mc)) return rc;
#if RRLW_GEN_BESIDE_SWEEP
            if (gen.on) HIP_TRY(hipEventRecord(G.ev_layer[k], s));
#endif
            if (int rc = run_sweep<true>(s, Wk, nb, col0, ncol, nlay, mode, idrv, 1, 16, g, c, out, mc)) return rc;
            HIP_TRY(hipEventRecord(G.ev_done[k], s));
        }
    }
    if (capture) {
        (void)aux_;
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(G.cap, &graph);
        s = s_call;
        if (rc_cap != 0) { if (graph) (void)hipGraphDestroy(graph); gent->failed = true; return rc_cap; }
        hipGraphExec_t exec = nullptr;
        if (ce != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            // no graph on this runtime for this call shape: the plain launches, now and from here on
            (void)hipGetLastError();
            if (graph) (void)hipGraphDestroy(graph);
            gent->failed = true;
            return run_pipelined(s, ncol, nlay, mode, idrv, g, inflag, iceflag, liqflag, out, mc, gen);
        }
        (void)hipGraphDestroy(graph);
        gent->exec = exec;
        G.graph_captures++;
        HIP_TRY(hipGraphLaunch(exec, s));
    }
    if (split) {                                  // the caller's stream sees the results of every batch
        for (int k = 0; k < std::min(i, 2); k++) HIP_TRY(hipStreamWaitEvent(s, G.ev_done[k], 0));
    }
    HIP_TRY(hipEventRecord(G.ev_last, s));
    G.ev_last_valid = true;
    return 0;
}

int read_physics_error(hipStream_t s)
{
    int code = 0;
    HIP_TRY(hipMemcpyAsync(&code, G.d_err, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (code != 0) {
        int zero = 0;
        HIP_TRY(hipMemcpy(G.d_err, &zero, sizeof(int), hipMemcpyHostToDevice));
        return fail(RRTMG_LW_HIP_EPHYSICS, "%s", physics_message(code));
    }
    return 0;
}

int ensure_stage(size_t bytes)
{
    if (G.stage_bytes >= bytes) return 0;
    if (G.stage_base) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(G.stage_base)); G.stage_base = nullptr; G.stage_bytes = 0; }
    HIP_TRY(hipMalloc(&G.stage_base, bytes));
    G.stage_bytes = bytes;
    return 0;
}

// the McICA entries exist in both builds: ngptlw sub-columns (src/rrtmg_lw_rad.f90:140, :267-288), masks of MASK_WORDS words
int check_mcica_build()
{
    return 0;
}

int check_common(int ncol, int nlay)
{
    if (!G.init) return fail(RRTMG_LW_HIP_ENOTINIT, "rrtmg_lw_hip_init has not been called");
    if (ncol < 1 || nlay < 1 || nlay > 603) return fail(RRTMG_LW_HIP_EARG, "bad dimensions ncol=%d nlay=%d", ncol, nlay);
    return 0;
}


// ---- host-pointer staging ---------------------------------------------------------------------------
// Every array of the interface is [rows][ncol][inner] with `inner` fastest (inner = 1 for (ncol,nlay) arrays,
// 16 for taucld, 140 for the McICA sub-column arrays); a column block is therefore one 2-D copy.
// The pipelined entries (host_pipeline) fill `src` per batch: where the batch's rows are read from - the caller's array, or an entry's
// own pinned scratch for an array it reduces on the host (tauctot).
struct HostIn {
    const double *h; size_t inner, rows; double *d;
    bool pinned = false;                                  // h is page-locked (rrtmg_lw_hip_host_register, or pinned by the caller's own means)
    const double *src = nullptr; size_t src_ncol = 0, src_col0 = 0; bool src_pinned = false;
    const unsigned char *skip = nullptr;                  // per batch, set by the entry's prep step: rows nothing reads (a cloud array's layers without cloud) - not scanned, zero-filled
    const unsigned char *known = nullptr;                 // ... rows the prep step has already read in full and found to hold one pattern (known_bits): not scanned again
    const uint64_t *known_bits = nullptr;
    unsigned long long static_gen = 0;                    // != 0: the caller declared the array static (rrtmg_lw_hip_host_static): row scans are cached per batch
};
struct HostOut { double *h; size_t rows; double *d; bool active; bool pinned = false; };
int bounce_h2d(void *dst, const void *src, size_t bytes);
int bounce_d2h(void *dst, const void *src, size_t bytes);
int bounce_h2d_rows(void *dst, const void *src, size_t row_bytes, size_t stride_bytes, size_t rows);

int stage_alloc(std::vector<HostIn> &ins, std::vector<HostOut> &outs, size_t nb)
{
    size_t tot = 0;
    for (auto &a : ins) tot += a.h ? a.inner * a.rows * nb : 0;
    for (auto &a : outs) tot += a.rows * nb;
    if (int rc = ensure_stage(tot * 8 + 4096)) return rc;
    double *p = (double *)G.stage_base;
    for (auto &a : ins) { a.d = a.h ? p : nullptr; p += a.h ? a.inner * a.rows * nb : 0; }
    for (auto &a : outs) { a.d = p; p += a.rows * nb; }
    return 0;
}

// the plain form (small one-batch entries: get_alpha, the stand-alone generator): blocking copies through the library's pinned buffer (bounce_h2d)
int stage_in(std::vector<HostIn> &ins, size_t ncol, size_t col0, size_t nb, hipStream_t s)
{
    HIP_TRY(hipStreamSynchronize(s));                    // (the staging buffer's last readers)
    for (auto &a : ins) {
        if (!a.h) continue;
        const size_t w = a.inner * nb * 8;
        if (nb == ncol) { if (int rc = bounce_h2d(a.d, a.h, w * a.rows)) return rc; continue; }
        if (int rc = bounce_h2d_rows(a.d, a.h + a.inner * col0, w, a.inner * ncol * 8, a.rows)) return rc;
    }
    return 0;
}

int stage_out(std::vector<HostOut> &outs, size_t ncol, size_t col0, size_t nb, hipStream_t s)
{
    HIP_TRY(hipStreamSynchronize(s));
    for (auto &a : outs) {
        if (!a.active || !a.h) continue;
        const size_t w = nb * 8;
        if (nb == ncol) { if (int rc = bounce_d2h(a.h, a.d, w * a.rows)) return rc; continue; }
        for (size_t r = 0; r < a.rows; r++)
            if (int rc = bounce_d2h(a.h + col0 + ncol * r, a.d + nb * r, w)) return rc;
    }
    return 0;
}

// f(t, nt) on nt host threads (the host-pointer entries' own work on the caller's arrays: row scans, packing, tauctot)
int host_threads()
{
    static const int nt = []() {         // (initialised once, thread-safe: the fan-out's worker threads all come through here)
        const char *e = getenv("RRTMG_LW_HOST_THREADS");
        int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
#ifdef __linux__
        cpu_set_t set;
        if (!e && sched_getaffinity(0, sizeof set, &set) == 0) v = std::min(v, CPU_COUNT(&set));
#endif
        return std::max(1, std::min(v, e ? 64 : 16));     // (sixteen unless the variable asks for more: the scan is bound by the memory of the NUMA node the arrays live on)
    }();
    return nt;
}
// The host threads persist (a batch of the host-pointer entries has three parallel regions of a millisecond or two each: fifteen thread
// starts per region cost a fifth of that).  One region at a time: a second caller - the worker thread of another device of the fan-out -
// starts its own threads instead of waiting.
class HostPool {
    std::mutex use_, mu_;
    std::condition_variable go_, done_;
    std::vector<std::thread> th_;
    const std::function<void(int, int)> *job_ = nullptr;
    int nt_ = 0, pending_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
    void worker(int t)
    {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int, int)> *job;
            int nt;
            {
                std::unique_lock<std::mutex> lk(mu_);
                go_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_; job = job_; nt = nt_;
            }
            if (t < nt) {
                (*job)(t, nt);
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
public:
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        go_.notify_all();
        for (auto &x : th_) x.join();
    }
    // f(t, nt) for t = 0 .. nt - 1 (t = 0 on the calling thread); false when the pool is in use
    bool run(int nt, const std::function<void(int, int)> &f)
    {
        std::unique_lock<std::mutex> use(use_, std::try_to_lock);
        if (!use.owns_lock()) return false;
        while ((int)th_.size() < nt - 1) { const int t = (int)th_.size() + 1; th_.emplace_back([this, t] { worker(t); }); }
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &f; nt_ = nt; pending_ = nt - 1; gen_++;
        }
        go_.notify_all();
        f(0, nt);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        return true;
    }
};
// One pool per device of the fan-out: the worker thread that feeds device d (fan_out) runs its parallel regions on pool d with its share
// of the host threads (host_threads() / devices: the scans are bound by the host's memory, more threads than cores only take turns).
HostPool g_pools[MAXDEV];
thread_local int tl_pool = 0, tl_share = 1;

template <class F>
void host_parallel(size_t work_bytes, F f)
{
    // (a thread per MB of host data at least: a call of a few columns does not wake sixteen threads)
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, host_threads() / tl_share), work_bytes >> 20));
    if (nt == 1) { f(0, 1); return; }
    const std::function<void(int, int)> fn = [&f](int t, int n) { f(t, n); };
    if (g_pools[tl_pool].run(nt, fn)) return;
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back([=, &f]() { f(t, nt); });
    f(0, nt);
    for (auto &x : th) x.join();
}

// flags[r] = 1 when every value of row r of the column batch [col0, col0 + nb) of a (rows, ncol, inner) array is below `thr` (a NaN is not).
// With thr = cldmin: a layer of the cloud fraction without cloud in any column of the batch - cldprop / cldprmc then read nothing else of that
// layer (src/rrtmg_lw_cldprop.f90:185-186, src/rrtmg_lw_cldprmc.f90:182-183), so the other cloud arrays' rows of that layer need not be read here.
// uni / bits (optional): rows that were read to their end (the ones below the threshold) and hold ONE 8-byte pattern - stage_rows need not read them again.
void rows_below(const double *h, size_t inner, size_t rows, size_t ncol, size_t col0, size_t nb, double thr, unsigned char *flags,
                unsigned char *uni = nullptr, uint64_t *bits = nullptr)
{
    host_parallel(inner * rows * nb * 8, [&](int t, int nt) {
        for (size_t r = (size_t)t; r < rows; r += (size_t)nt) {
            const double *p = h + inner * (col0 + ncol * r);
            const uint64_t *u = reinterpret_cast<const uint64_t *>(p);
            const size_t n = inner * nb;
            const uint64_t v = u[0];
            uint64_t acc = 0;
            bool below = true;
            size_t i = 0;
            for (; i + 64 <= n && below; i += 64) {
                int ok = 1;
                for (size_t e = 0; e < 64; e++) { ok &= (int)(p[i + e] < thr); acc |= u[i + e] ^ v; }
                below = ok != 0;
            }
            for (; i < n && below; i++) { below = p[i] < thr; acc |= u[i] ^ v; }
            flags[r] = below ? 1 : 0;
            if (uni) { uni[r] = (below && acc == 0) ? 1 : 0; bits[r] = v; }
        }
    });
}

// Ranges pinned through rrtmg_lw_hip_host_register (and the chunk queue's own pinned set): an array lies in one of them with all its bytes,
// or it is treated as pageable and goes through the pinned staging.  (The runtime's pointer attributes were asked as well, at an array's
// first and last byte, until tools/soak_host_entry.py found the hole: registrations are page-granular, so an UNREGISTERED array whose two
// ends share pages with registered neighbours - small arrays side by side on the heap - looked pinned, its middle pages were not mapped for
// the device, and the direct copy faulted.  An array pinned by other means than this library's call is copied like a pageable one.)
std::vector<std::pair<const char *, size_t>> g_pinned;
void forget_pinned(const void *ptr)
{
    for (size_t i = 0; i < g_pinned.size(); i++)
        if (g_pinned[i].first == (const char *)ptr) { g_pinned.erase(g_pinned.begin() + (long)i); break; }
}
bool host_range_pinned(const void *p, size_t bytes)
{
    const char *a = (const char *)p;
    if (!a || bytes == 0) return false;
    for (auto &r : g_pinned)
        if (a >= r.first && a + bytes <= r.first + r.second) return true;
    return false;
}

// Arrays the caller has declared static (rrtmg_lw_hip_host_static): their contents stay as they are until rrtmg_lw_hip_host_changed.
// The host-pointer entries scan every input row of every call for "one value for all columns of the batch" (stage_rows) - 14 KB per
// 72-layer column of reads that bound the entry once the arrays are pinned; for a static array (well-mixed gases handed over as full
// arrays, zero aerosol, emissivities) the answer of the first call is kept per (array, column batch) and the rows are not read again:
// uniform rows are filled on the device as before, the others go straight to their DMA.
struct StaticRange { const char *p; size_t bytes; unsigned long long gen; };
std::vector<StaticRange> g_static;
unsigned long long g_static_gen = 1;
struct ScanKey {
    const void *base; size_t ncol, inner, rows, col0, nb;
    bool operator<(const ScanKey &o) const
    {
        if (base != o.base) return base < o.base;
        if (ncol != o.ncol) return ncol < o.ncol;
        if (inner != o.inner) return inner < o.inner;
        if (rows != o.rows) return rows < o.rows;
        if (col0 != o.col0) return col0 < o.col0;
        return nb < o.nb;
    }
};
struct ScanVal { unsigned long long gen = 0; std::vector<unsigned char> state; std::vector<uint64_t> bits; };     // state: 0 not scanned, 1 uniform, 2 not uniform
std::map<ScanKey, ScanVal> g_scan;
std::mutex g_scan_mu;          // (the fan-out's worker threads stage their batches side by side)
// generation of the static range that holds the whole array, 0 if none
unsigned long long static_gen_of(const void *p, size_t bytes)
{
    const char *a = (const char *)p;
    for (auto &r : g_static)
        if (a >= r.p && a + bytes <= r.p + r.bytes) return r.gen;
    return 0;
}

int ensure_hostbuf(char **p, size_t *cap, size_t bytes)
{
    if (*cap >= bytes) return 0;
    if (*p) { HIP_TRY(hipHostFree(*p)); *p = nullptr; *cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipHostMalloc((void **)p, want, hipHostMallocDefault));
    *cap = want;
    return 0;
}

// Blocking copies between the caller's (possibly pageable) memory and the device through this library's own pinned buffer: the runtime is
// never handed a pageable user pointer.  It would pin the range in place, page-granular, and unpin it afterwards - and the unpinning takes a
// page shared with a neighbouring array out of the device's page table as well, whether that neighbour is the source of another copy in
// flight or an array the caller registered (tools/soak_host_entry.py: "Memory access fault" at heap addresses).
constexpr size_t BOUNCE_BYTES = (size_t)32 << 20;
int bounce_h2d(void *dst, const void *src, size_t bytes)
{
    auto &hs = G.hset[0];
    if (int rc = ensure_hostbuf(&hs.in, &hs.in_cap, std::min(bytes, BOUNCE_BYTES))) return rc;
    for (size_t o = 0; o < bytes; o += BOUNCE_BYTES) {
        const size_t n = std::min(BOUNCE_BYTES, bytes - o);
        host_parallel(2 * n, [&](int t, int nt) {
            const size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt;
            memcpy(hs.in + a, (const char *)src + o + a, b - a);
        });
        HIP_TRY(hipMemcpy((char *)dst + o, hs.in, n, hipMemcpyHostToDevice));
    }
    return 0;
}
int bounce_d2h(void *dst, const void *src, size_t bytes)
{
    auto &hs = G.hset[0];
    if (int rc = ensure_hostbuf(&hs.out, &hs.out_cap, std::min(bytes, BOUNCE_BYTES))) return rc;
    for (size_t o = 0; o < bytes; o += BOUNCE_BYTES) {
        const size_t n = std::min(BOUNCE_BYTES, bytes - o);
        HIP_TRY(hipMemcpy(hs.out, (const char *)src + o, n, hipMemcpyDeviceToHost));
        host_parallel(2 * n, [&](int t, int nt) {
            const size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt;
            memcpy((char *)dst + o + a, hs.out + a, b - a);
        });
    }
    return 0;
}

// `rows` rows of `row_bytes` each, `stride_bytes` apart in the caller's memory, to a contiguous device buffer (a column block of a
// (ncol, nlay) array), packed into the pinned buffer by the host threads first
int bounce_h2d_rows(void *dst, const void *src, size_t row_bytes, size_t stride_bytes, size_t rows)
{
    if (row_bytes == stride_bytes) return bounce_h2d(dst, src, row_bytes * rows);
    auto &hs = G.hset[0];
    const size_t per = std::max<size_t>(1, BOUNCE_BYTES / row_bytes);
    if (int rc = ensure_hostbuf(&hs.in, &hs.in_cap, std::min(rows, per) * row_bytes)) return rc;
    for (size_t r0 = 0; r0 < rows; r0 += per) {
        const size_t nr = std::min(per, rows - r0);
        host_parallel(2 * nr * row_bytes, [&](int t, int nt) {
            for (size_t r = nr * (size_t)t / (size_t)nt; r < nr * (size_t)(t + 1) / (size_t)nt; r++)
                memcpy(hs.in + r * row_bytes, (const char *)src + (r0 + r) * stride_bytes, row_bytes);
        });
        HIP_TRY(hipMemcpy((char *)dst + r0 * row_bytes, hs.in, nr * row_bytes, hipMemcpyHostToDevice));
    }
    return 0;
}

int ensure_copy_streams()
{
    if (G.cp_in) return 0;
    HIP_TRY(hipStreamCreateWithFlags(&G.cp_in, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&G.cp_out, hipStreamNonBlocking));
    for (int k = 0; k < HOST_SETS; k++) {
        HIP_TRY(hipEventCreateWithFlags(&G.ev_h2d[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&G.ev_cmp[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&G.ev_d2h[k], hipEventDisableTiming));
    }
    return 0;
}

// One column batch of the input arrays -> device staging (stream s), moving only what has to move:
//  * a row (one layer, or one (layer, band) of tauaer) whose nb x inner values all hold the same 8-byte pattern does not travel: runs of
//    such rows become entries of the k_fill_rows table.  Well-mixed gases handed over as constants, cloud arrays outside the cloudy
//    layers and aerosol optical depths outside the aerosol layers are most of a GCM call's bytes; the device arrays are bit-identical
//    to copied ones.  A row that is not uniform leaves the scan at its first 64 values.
//  * rows of a PAGEABLE array are packed into the pinned set by the host threads and leave as one asynchronous DMA per run of rows
//    (the runtime's own pageable path is one blocking, single-threaded staging copy per call); rows of a pinned array leave from
//    where they lie.
// The pinned set k is free: the caller has waited for the DMAs of the batch that used it last.
struct StagedRow { unsigned arr, row; bool uniform; uint64_t bits; size_t off; };
int stage_rows(std::vector<HostIn> &ins, size_t nb, int k, hipStream_t s)
{
    static thread_local std::vector<StagedRow> tl_rows;
    std::vector<StagedRow> &rows = tl_rows;          // (the host threads below must see THIS thread's list)
    rows.clear();
    size_t bytes = 0, nrows = 0;
    for (auto &a : ins) nrows += a.h ? a.rows : 0;
    rows.reserve(nrows);
    for (size_t ai = 0; ai < ins.size(); ai++) {
        const HostIn &a = ins[ai];
        if (!a.h) continue;
        for (size_t r = 0; r < a.rows; r++) rows.push_back({(unsigned)ai, (unsigned)r, false, 0, 0});
        bytes += a.inner * a.rows * nb * 8;
    }
    auto row_src = [&](const StagedRow &q) { const HostIn &a = ins[q.arr]; return a.src + a.inner * (a.src_col0 + a.src_ncol * q.row); };
    // cached scans of the arrays declared static (the entry of this array and batch; filled below on a miss)
    std::vector<ScanVal *> cache(ins.size(), nullptr);
    std::vector<unsigned char> cache_hit(ins.size(), 0);
    {
        std::lock_guard<std::mutex> lk(g_scan_mu);
        for (size_t ai = 0; ai < ins.size(); ai++) {
            const HostIn &a = ins[ai];
            if (!a.h || a.static_gen == 0 || a.src != a.h) continue;          // (an array the entry reduces on the host, tauctot, is formed anew per call)
            ScanVal &v = g_scan[ScanKey{a.h, a.src_ncol, a.inner, a.rows, a.src_col0, nb}];
            if (v.gen == a.static_gen && v.state.size() == a.rows) cache_hit[ai] = 1;
            else { v.gen = a.static_gen; v.state.assign(a.rows, 0); v.bits.assign(a.rows, 0); }
            cache[ai] = &v;                                                     // (map nodes do not move; entries are dropped under the entry lock only)
        }
    }
    host_parallel(bytes, [&](int t, int nt) {
        for (size_t j = (size_t)t; j < rows.size(); j += (size_t)nt) {
            StagedRow &q = rows[j];
            if (ins[q.arr].skip && ins[q.arr].skip[q.row]) { q.uniform = true; q.bits = 0; continue; }
            if (ins[q.arr].known && ins[q.arr].known[q.row]) { q.uniform = true; q.bits = ins[q.arr].known_bits[q.row]; continue; }
            if (cache_hit[q.arr] && cache[q.arr]->state[q.row] != 0) { q.uniform = cache[q.arr]->state[q.row] == 1; q.bits = cache[q.arr]->bits[q.row]; continue; }
            const uint64_t *p = reinterpret_cast<const uint64_t *>(row_src(q));
            const size_t n = ins[q.arr].inner * nb;
            const uint64_t v = p[0];
            uint64_t acc = 0;
            size_t i = 0;
            for (; i + 64 <= n && acc == 0; i += 64)
                for (size_t e = 0; e < 64; e++) acc |= p[i + e] ^ v;
            for (; i < n && acc == 0; i++) acc |= p[i] ^ v;
            q.uniform = acc == 0;
            q.bits = v;
            if (cache[q.arr]) { cache[q.arr]->bits[q.row] = v; cache[q.arr]->state[q.row] = q.uniform ? 1 : 2; }      // (one thread per row)
        }
    });
    size_t need = 0;
    for (auto &q : rows)
        if (!q.uniform && !ins[q.arr].src_pinned) { q.off = need; need += ins[q.arr].inner * nb * 8; }
    auto &hs = G.hset[k];
    if (int rc = ensure_hostbuf(&hs.in, &hs.in_cap, need)) return rc;
    if (need)
        host_parallel(2 * need, [&](int t, int nt) {
            for (size_t j = (size_t)t; j < rows.size(); j += (size_t)nt) {
                const StagedRow &q = rows[j];
                if (!q.uniform && !ins[q.arr].src_pinned) memcpy(hs.in + q.off, row_src(q), ins[q.arr].inner * nb * 8);
            }
        });
    RowFill *tab = reinterpret_cast<RowFill *>(hs.fill);
    size_t nf = 0;
    for (size_t j = 0; j < rows.size();) {
        const StagedRow &q = rows[j];
        const HostIn &a = ins[q.arr];
        const size_t w = a.inner * nb;                    // values per row
        size_t j1 = j + 1;
        while (j1 < rows.size() && rows[j1].arr == q.arr && rows[j1].uniform == q.uniform && (!q.uniform || rows[j1].bits == q.bits)) j1++;
        const size_t nr = j1 - j;
        double *dst = a.d + w * q.row;
        if (q.uniform) {
            for (size_t done = 0; done < nr * w; done += FILL_MAX)
                tab[nf++] = RowFill{reinterpret_cast<unsigned long long *>(dst + done), std::min<unsigned long long>(FILL_MAX, nr * w - done), q.bits};
        } else if (a.src_pinned) {
            HIP_TRY(hipMemcpy2DAsync(dst, w * 8, row_src(q), a.inner * a.src_ncol * 8, w * 8, nr, hipMemcpyHostToDevice, s));
        } else {
            HIP_TRY(hipMemcpyAsync(dst, hs.in + q.off, nr * w * 8, hipMemcpyHostToDevice, s));
        }
        j = j1;
    }
    for (size_t f0 = 0; f0 < nf; f0 += 65535)
        hipLaunchKernelGGL(k_fill_rows, dim3(FILL_BLOCKS, (unsigned)std::min<size_t>(65535, nf - f0)), dim3(256), 0, s, (const RowFill *)(tab + f0));
    return 0;
}

// Host-pointer entries as a three-stage pipeline over the column batches: H2D of batch i+1 (copy stream) | kernels of
// batch i (G.stream) | D2H of batch i-1 (second copy stream), HOST_SETS staging sets on the device and as many pinned sets on the host.
// The calling thread (with the host threads) scans and packs batch i+1 and unpacks the outputs of batch i-1 while the DMAs and
// kernels of the batches between are in flight; nothing it calls blocks on a copy.  body(stream, nb, col0, in, out) enqueues the
// kernels of one batch whose staged arrays start at column 0; prep(k, col0, nb, stream) is the entry's own host work for the batch.
// RRTMG_LW_STAGE_TIMING=1: per call, the pipeline prints where its time went (host phases by the clock, device stages by events)
const bool g_stage_timing = []() { const char *e = getenv("RRTMG_LW_STAGE_TIMING"); return e && atoi(e) != 0; }();
struct StageClock {
    double t[8] = {};          // 0 wait h2d, 1 unpack (incl. wait for the D2H), 2 prep, 3 scan + pack + enqueue, 4 body enqueue, 5 copy_out enqueue, 6 final drain
    std::chrono::steady_clock::time_point last;
    void start() { last = std::chrono::steady_clock::now(); }
    void lap(int k) { auto n = std::chrono::steady_clock::now(); t[k] += std::chrono::duration<double, std::milli>(n - last).count(); last = n; }
};
struct NoPrep { int operator()(int, int, int, hipStream_t) const { return 0; } };
template <class Body, class Prep>
int host_pipeline_run(int ncol, int c0, int c1, int nbmax, std::vector<HostIn> &ins, std::vector<HostOut> &outs, Body body, Prep prep, std::vector<hipEvent_t> &tev);
template <class Body, class Prep = NoPrep>
int host_pipeline(int ncol, int c0, int c1, int nbmax, std::vector<HostIn> &ins, std::vector<HostOut> &outs, Body body, Prep prep = Prep())
{
    std::vector<hipEvent_t> tev;             // (timing) per batch: H2D begin / end, kernels begin / end
    const int rc = host_pipeline_run(ncol, c0, c1, nbmax, ins, outs, body, prep, tev);
    if (rc != 0) {
        // An error part-way leaves up to HOST_SETS batches of DMAs and kernels in flight, some of them to and from the caller's registered
        // arrays: nothing of this call may still be running when the caller gets its error (and frees or unregisters them).
        if (G.cp_in) (void)hipStreamSynchronize(G.cp_in);
        if (G.stream) (void)hipStreamSynchronize(G.stream);
        if (G.cp_out) (void)hipStreamSynchronize(G.cp_out);
    }
    for (auto e : tev) (void)hipEventDestroy(e);
    return rc;
}
template <class Body, class Prep>
int host_pipeline_run(int ncol, int c0, int c1, int nbmax, std::vector<HostIn> &ins, std::vector<HostOut> &outs, Body body, Prep prep, std::vector<hipEvent_t> &tev)
{   // columns [c0, c1) of arrays that are ncol columns wide
    if (int rc = ensure_copy_streams()) return rc;
    size_t set = 0, nrows = 0, out_pageable = 0;
    for (auto &a : ins) {
        set += a.h ? a.inner * a.rows * (size_t)nbmax : 0;
        nrows += a.h ? a.rows + a.inner * a.rows * (size_t)nbmax / FILL_MAX + 1 : 0;
        a.pinned = a.h && host_range_pinned(a.h, a.inner * (size_t)ncol * a.rows * 8);
        a.static_gen = a.h ? static_gen_of(a.h, a.inner * (size_t)ncol * a.rows * 8) : 0;
    }
    for (auto &a : outs) {
        set += a.rows * (size_t)nbmax;
        a.pinned = a.active && a.h && host_range_pinned(a.h, (size_t)ncol * a.rows * 8);
        if (a.active && a.h && !a.pinned) out_pageable += a.rows * (size_t)nbmax * 8;
    }
    set = align_up(set * 8, 256);
    if (int rc = ensure_stage(HOST_SETS * set + 4096)) return rc;
    for (auto &hs : G.hset) {
        if (int rc = ensure_hostbuf(&hs.fill, &hs.fill_cap, nrows * sizeof(RowFill))) return rc;
        if (int rc = ensure_hostbuf(&hs.out, &hs.out_cap, out_pageable)) return rc;
    }
    auto bind = [&](int k) {                 // point the descriptors at staging set k
        double *p = (double *)((char *)G.stage_base + (size_t)k * set);
        for (auto &a : ins) { a.d = a.h ? p : nullptr; p += a.h ? a.inner * a.rows * (size_t)nbmax : 0; }
        for (auto &a : outs) { a.d = p; p += a.rows * (size_t)nbmax; }
    };
    struct Pending { bool on = false; int col0 = 0, nb = 0; } pend[HOST_SETS];
    auto copy_out = [&](int k, int col0, int nb) -> int {
        bind(k);
        HIP_TRY(hipStreamWaitEvent(G.cp_out, G.ev_cmp[k], 0));
        char *p = G.hset[k].out;
        bool any = false;
        for (auto &a : outs) {
            if (!a.active || !a.h) continue;
            const size_t w = (size_t)nb * 8, sp = (size_t)ncol * 8;
            if (a.pinned) { HIP_TRY(hipMemcpy2DAsync(a.h + col0, sp, a.d, w, w, a.rows, hipMemcpyDeviceToHost, G.cp_out)); continue; }
            HIP_TRY(hipMemcpyAsync(p, a.d, a.rows * w, hipMemcpyDeviceToHost, G.cp_out));
            p += a.rows * w;
            any = true;
        }
        HIP_TRY(hipEventRecord(G.ev_d2h[k], G.cp_out));
        pend[k] = Pending{any, col0, nb};
        return 0;
    };
    // the pageable output arrays' rows of the batch that used set k last: pinned set -> the caller's arrays.  must = false: only if its D2H has landed
    auto unpack = [&](int k, bool must) -> int {
        if (!pend[k].on) return 0;
        if (!must) {
            const hipError_t q = hipEventQuery(G.ev_d2h[k]);
            if (q == hipErrorNotReady) return 0;
            if (q != hipSuccess) return fail(RRTMG_LW_HIP_EHIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        }
        pend[k].on = false;
        HIP_TRY(hipEventSynchronize(G.ev_d2h[k]));
        const size_t nb = (size_t)pend[k].nb, col0 = (size_t)pend[k].col0;
        struct Seg { double *dst; const double *src; size_t rows; };
        std::vector<Seg> segs;
        size_t tot = 0;
        const char *p = G.hset[k].out;
        for (auto &a : outs) {
            if (!a.active || !a.h || a.pinned) continue;
            segs.push_back({a.h + col0, (const double *)p, a.rows});
            p += a.rows * nb * 8;
            tot += a.rows;
        }
        host_parallel(2 * tot * nb * 8, [&](int t, int nt) {
            const size_t r0 = tot * (size_t)t / (size_t)nt, r1 = tot * (size_t)(t + 1) / (size_t)nt;
            size_t base = 0;
            for (auto &sg : segs) {
                for (size_t r = std::max(r0, base); r < std::min(r1, base + sg.rows); r++)
                    memcpy(sg.dst + (size_t)ncol * (r - base), sg.src + nb * (r - base), nb * 8);
                base += sg.rows;
            }
        });
        return 0;
    };
    StageClock clk;
    auto tmark = [&](hipStream_t st) { if (!g_stage_timing) return; hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, st); tev.push_back(e); };
    const auto t_call = std::chrono::steady_clock::now();
    int i = 0, prev_col0 = 0, prev_nb = 0;
    for (int col0 = c0; col0 < c1; col0 += nbmax, i++) {
        const int nb = std::min(nbmax, c1 - col0), k = i % HOST_SETS, kprev = (i + HOST_SETS - 1) % HOST_SETS;
        bind(k);
        clk.start();
        if (i >= HOST_SETS) {
            HIP_TRY(hipStreamWaitEvent(G.cp_in, G.ev_cmp[k], 0));          // kernels of batch i - HOST_SETS have read staging set k
            HIP_TRY(hipEventSynchronize(G.ev_h2d[k]));                     // its DMAs have left the pinned set k (this thread runs ahead of the device)
            clk.lap(0);
            if (int rc = unpack(k, true)) return rc;
            clk.lap(1);
        }
        for (auto &a : ins) { a.src = a.h; a.src_ncol = (size_t)ncol; a.src_col0 = (size_t)col0; a.src_pinned = a.pinned; a.skip = nullptr; a.known = nullptr; a.known_bits = nullptr; }
        // the entry's own host work for this batch (reductions into its pinned scratch set k, which it then names as an array's source)
        if (int rc = prep(k, col0, nb, G.cp_in)) return rc;
        clk.lap(2);
        tmark(G.cp_in);
        if (int rc = stage_rows(ins, (size_t)nb, k, G.cp_in)) return rc;
        tmark(G.cp_in);
        HIP_TRY(hipEventRecord(G.ev_h2d[k], G.cp_in));
        clk.lap(3);
        HIP_TRY(hipStreamWaitEvent(G.stream, G.ev_h2d[k], 0));
        if (i >= HOST_SETS) HIP_TRY(hipStreamWaitEvent(G.stream, G.ev_d2h[k], 0));         // outputs of batch i - HOST_SETS have left staging set k
        tmark(G.stream);
        if (int rc = body(G.stream, nb, col0, ins, outs)) return rc;
        tmark(G.stream);
        HIP_TRY(hipEventRecord(G.ev_cmp[k], G.stream));
        clk.lap(4);
        if (i >= 1)
            if (int rc = copy_out(kprev, prev_col0, prev_nb)) return rc;
        clk.lap(5);
        // outputs that have landed in the meantime (oldest first): unpacked now, while the device works on the batches just enqueued
        for (int j = 1; j < HOST_SETS; j++)
            if (int rc = unpack((k + j) % HOST_SETS, false)) return rc;
        clk.lap(1);
        prev_col0 = col0; prev_nb = nb;
    }
    clk.start();
    if (int rc = copy_out((i + HOST_SETS - 1) % HOST_SETS, prev_col0, prev_nb)) return rc;
    for (int j = 0; j < HOST_SETS; j++)                  // (oldest first)
        if (int rc = unpack((i + j) % HOST_SETS, true)) return rc;
    HIP_TRY(hipStreamSynchronize(G.cp_out));
    HIP_TRY(hipStreamSynchronize(G.stream));
    clk.lap(6);
    if (g_stage_timing) {
        double h2d = 0.0, ker = 0.0;
        for (size_t e = 0; e + 3 < tev.size(); e += 4) {
            float a = 0.f, b = 0.f;
            (void)hipEventElapsedTime(&a, tev[e], tev[e + 1]);
            (void)hipEventElapsedTime(&b, tev[e + 2], tev[e + 3]);
            h2d += a; ker += b;
        }
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
        fprintf(stderr, "[rrtmg_lw_hip stage] %d columns in %d batches of <= %d: wall %.2f ms | host: wait-h2d %.2f unpack(+wait d2h) %.2f prep %.2f scan+pack+enqueue %.2f "
                        "body-enqueue %.2f copy_out-enqueue %.2f drain %.2f | device: H2D stream busy %.2f kernels %.2f\n",
                c1 - c0, i, nbmax, wall, clk.t[0], clk.t[1], clk.t[2], clk.t[3], clk.t[4], clk.t[5], clk.t[6], h2d, ker);
    }
    return 0;
}

// ---- Mersenne Twister MT19937 (Matsumoto & Nishimura 1998; init_genrand / genrand_real1 of mt19937ar) -------
// The reference's irng = 1 stream: src/mcica_random_numbers.f90:157-169 (scalar seeding), :262-295 (deviate on [0,1]).
struct MT19937 {
    uint32_t st[624];
    int cur = 624;
    explicit MT19937(uint32_t seed)
    {
        st[0] = seed;
        for (int i = 1; i < 624; i++) st[i] = 1812433253u * (st[i - 1] ^ (st[i - 1] >> 30)) + (uint32_t)i;
    }
    void refill()
    {
        for (int k = 0; k < 624; k++) {
            const uint32_t y = (st[k] & 0x80000000u) | (st[(k + 1) % 624] & 0x7fffffffu);
            st[k] = st[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        cur = 0;
    }
    double real1()
    {
        if (cur >= 624) refill();
        uint32_t y = st[cur++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return (double)y / 4294967295.0;
    }
};

// ---- chunk-start states of the Mersenne-Twister stream (k_mt_jump, mtjump.hpp) ---------------------------------------------------
// The stream of a call is NGPT slabs of `per` deviates (one sub-column each), every slab cut into M chunks of C: state (isub, m) is the
// MT19937 state isub per + m C deviates into the stream.  Two levels of doubling rounds seat them: slab starts (x^(per 2^k) applied to
// the 2^k slabs that exist), then chunk starts inside all slabs at once (x^(C 2^k)).  Cached per (seed, per), two sets (a host model that
// keeps its permuteseed pays for the jumps once; one that hands its columns over in blocks with a shorter last block alternates between
// two values of `per`).
struct MtStates {
    long long seed = -1;
    unsigned long long per = 0, C = 0;
    int M = 0;
    unsigned *dev = nullptr;            // [NGPT * M][624]
    size_t cap = 0;                     // states allocated
    unsigned long long *polys = nullptr;   // [32][MT_PW]
    unsigned long long used = 0;        // last use (the older set makes room)
    size_t bytes() const { return (dev ? cap * (size_t)MT_NW * sizeof(unsigned) : 0) + (polys ? (size_t)32 * MT_PW * sizeof(unsigned long long) : 0); }
};
MtStates g_mt[2];
mtj::Poly g_mt_phi{};
bool g_mt_have_phi = false;
unsigned long long g_mt_clock = 0;

int mt_states(hipStream_t s, uint32_t seed, unsigned long long per, int *M_out, unsigned long long *C_out, unsigned **dev_out)
{
    const int M = (int)std::min<unsigned long long>(512ull, std::max<unsigned long long>(1ull, (per + 65535ull) / 65536ull));   // a wavefront per chunk
    const unsigned long long C = (per + (unsigned long long)M - 1ull) / (unsigned long long)M;
    *M_out = M; *C_out = C;
    for (MtStates &S : g_mt)
        if (S.dev && S.seed == (long long)seed && S.per == per && S.M == M) { S.used = ++g_mt_clock; *dev_out = S.dev; return 0; }
    MtStates &T = g_mt[0].used <= g_mt[1].used ? g_mt[0] : g_mt[1];
    T.used = ++g_mt_clock;
    HIP_TRY(hipDeviceSynchronize());                                // an earlier call may still be reading the states
    if (!g_mt_have_phi) {
        g_mt_phi = mtj::char_poly();
        if (!mtj::bit(g_mt_phi.data(), mtj::DEG)) return fail(RRTMG_LW_HIP_EHIP, "MT19937: characteristic polynomial not found");
        g_mt_have_phi = true;
    }
    const mtj::Poly &phi = g_mt_phi;
    const size_t need = (size_t)NGPT * M;
    if (T.cap < need) {
        if (T.dev) HIP_TRY(hipFree(T.dev));
        T.dev = nullptr; T.cap = 0;
        HIP_TRY(hipMalloc((void **)&T.dev, need * MT_NW * sizeof(unsigned)));
        T.cap = need;
    }
    if (!T.polys) HIP_TRY(hipMalloc((void **)&T.polys, (size_t)32 * MT_PW * sizeof(unsigned long long)));
    T.seed = -1;
    // polynomials: slab level k = 0 .. K1-1 (2^K1 >= NGPT), chunk level k = 0 .. K2-1 (2^K2 >= M)
    int K1 = 0, K2 = 0;
    while ((1 << K1) < NGPT) K1++;
    while ((1 << K2) < M) K2++;
    std::vector<unsigned long long> host((size_t)(K1 + K2) * MT_PW);
    {
        mtj::Poly p = mtj::pow_x(per, phi);
        for (int k = 0; k < K1; k++) { std::memcpy(&host[(size_t)k * MT_PW], p.data(), sizeof(p)); p = mtj::sqr_mod(p, phi); }
        if (K2) {
            p = mtj::pow_x(C, phi);
            for (int k = 0; k < K2; k++) { std::memcpy(&host[(size_t)(K1 + k) * MT_PW], p.data(), sizeof(p)); p = mtj::sqr_mod(p, phi); }
        }
    }
    HIP_TRY(hipMemcpy(T.polys, host.data(), host.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    {
        MT19937 mt(seed);                                           // init_genrand (src/mcica_random_numbers.f90:157-169)
        HIP_TRY(hipMemcpy(T.dev, mt.st, sizeof(mt.st), hipMemcpyHostToDevice));
    }
    for (int k = 0; k < K1; k++) {
        const int have = 1 << k, n = std::min(have, NGPT - have);
        if (n <= 0) break;
        hipLaunchKernelGGL(k_mt_jump, dim3(n, 1), dim3(MT_BLOCK), 0, s, T.dev, (const unsigned long long *)(T.polys + (size_t)k * MT_PW), 0, M, 0, have * M);
    }
    for (int k = 0; k < K2; k++) {
        const int have = 1 << k, n = std::min(have, M - have);
        if (n <= 0) break;
        hipLaunchKernelGGL(k_mt_jump, dim3(n, NGPT), dim3(MT_BLOCK), 0, s, T.dev, (const unsigned long long *)(T.polys + (size_t)(K1 + k) * MT_PW), 0, 1, M, have);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RRTMG_LW_HIP_EHIP, "k_mt_jump launch failed: %s", hipGetErrorString(e));
    T.seed = (long long)seed; T.per = per; T.C = C; T.M = M;
    *dev_out = T.dev;
    return 0;
}

// the table of one (draws per sub-column, permuteseed) pair: entries 0 .. KJ_NGROUP-1 jump from the seed to sub-column 8 g, the last one
// by one sub-column.  One table, rebuilt (after a device synchronisation: an earlier call may still read it) when the pair changes - a host
// model keeps its permuteseed per call site, so in steady state this is a comparison.
// (struct KissTable: beside State, which holds one per device)

int kiss_table(hipStream_t s, int stride, int permuteseed, const KissJump **dev, KissJump *jsub)
{
    KissTable &T = G.kiss;
    const long long seed = std::max(permuteseed, 0);
    if (!T.dev) HIP_TRY(hipMalloc((void **)&T.dev, sizeof(T.host)));
    if (T.stride != (unsigned long long)stride || T.seed != seed) {
        // an earlier launch on another stream may still be reading the table
        HIP_TRY(hipDeviceSynchronize());
        for (int g = 0; g < KJ_NGROUP; g++) T.host[g] = kiss_jump_entry((unsigned long long)seed + 8ull * g * (unsigned long long)stride);
        T.host[KJ_NGROUP] = kiss_jump_entry((unsigned long long)stride);
        T.stride = (unsigned long long)stride; T.seed = seed;
        HIP_TRY(hipMemcpy(T.dev, T.host, sizeof(T.host), hipMemcpyHostToDevice));
    }
    (void)s;
    *dev = T.dev;
    *jsub = T.host[KJ_NGROUP];
    return 0;
}

template <int RULE>
void launch_kiss_rule(hipStream_t s, const Workspace &Wk, const SubcolIn &in, const KissJump *jt, const KissJump &jsub, int ncol, int col0, int nb, int nlay)
{
    const size_t lds = (size_t)nlay * KJ_COLS * sizeof(int4);
    const dim3 grid((nb + KJ_COLS - 1) / KJ_COLS), block(KJ_BLOCK);
    hipLaunchKernelGGL(k_subcol_kiss<RULE>, grid, block, lds, s, Wk, in, jt, jsub, ncol, col0, nb, nlay);
}

// kissvec masks of columns col0 .. col0+nb-1 (of ncol) on stream s; G.W.mask must be set up (prepare_mask)
int launch_kiss(hipStream_t s, const Workspace &Wk, int ncol, int col0, int nb, int nlay, int icld, int permuteseed, const SubcolIn &in)
{
    if (icld < 1 || icld > 5) return fail(RRTMG_LW_HIP_EARG, "the sub-column generator needs icld 1..5");
    const int stride = icld == 3 ? 1 : ((icld == 4 || icld == 5) ? 2 * nlay : nlay);    // draws per sub-column (:475-530)
    const KissJump *jt = nullptr;
    KissJump jsub;
    if (int rc = kiss_table(s, stride, permuteseed, &jt, &jsub)) return rc;
    State::ProfRec r{"k_subcol_kiss", nullptr, nullptr};
    if (G.profile) { r.a = get_event(); r.b = get_event(); (void)hipEventRecord(r.a, s); }
    switch (icld) {
    case 1: launch_kiss_rule<1>(s, Wk, in, jt, jsub, ncol, col0, nb, nlay); break;
    case 2: launch_kiss_rule<2>(s, Wk, in, jt, jsub, ncol, col0, nb, nlay); break;
    case 3: launch_kiss_rule<3>(s, Wk, in, jt, jsub, ncol, col0, nb, nlay); break;
    default: launch_kiss_rule<4>(s, Wk, in, jt, jsub, ncol, col0, nb, nlay); break;
    }
    if (G.profile) { (void)hipEventRecord(r.b, s); G.prof.push_back(r); }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RRTMG_LW_HIP_EHIP, "k_subcol_kiss launch failed: %s", hipGetErrorString(e));
    return 0;
}

// mask buffer for all `ncol` columns of the call + argument checks of the generator
int prepare_mask(int ncol, int nlay, int icld, int irng, const double *alpha)
{
    if (nlay < 4 && irng == 0) return fail(RRTMG_LW_HIP_EARG, "the kissvec generator needs at least four layers");
    if ((icld == 4 || icld == 5) && !alpha) return fail(RRTMG_LW_HIP_EARG, "icld = 4/5 needs alpha");
    if (int rc = ensure_mask(nlay, (size_t)ncol)) return rc;
    G.W.mask = G.mask;
    G.W.mask_stride = (size_t)ncol;
    G.W.mask_col0 = 0;
    G.W.err = G.d_err;
    if (irng == 0) {
        const size_t lds = (size_t)nlay * KJ_COLS * sizeof(int4);
        // (+ 64: the kernel's few bytes of static LDS count against the same limits)
        if (lds + 64 > 160 * 1024) return fail(RRTMG_LW_HIP_EARG, "nlay=%d exceeds the generator's LDS budget", nlay);
        if (lds + 64 > 48 * 1024) {
            HIP_TRY(hipFuncSetAttribute((const void *)k_subcol_kiss<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_TRY(hipFuncSetAttribute((const void *)k_subcol_kiss<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_TRY(hipFuncSetAttribute((const void *)k_subcol_kiss<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_TRY(hipFuncSetAttribute((const void *)k_subcol_kiss<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
    }
    return 0;
}

// Sub-column masks of all `ncol` columns into G.mask (device arrays play, cldfrac, alpha are (ncol,nlay)).
int generate_mask(hipStream_t s, int ncol, int nlay, int icld, int permuteseed, int irng, const double *play,
                  const double *cldfrac, const double *alpha)
{
    if (int rc = prepare_mask(ncol, nlay, icld, irng, alpha)) return rc;
    // the mask buffer is shared with an earlier device-entry call that may still be running on another stream
    if (G.ev_last_valid) HIP_TRY(hipStreamWaitEvent(s, G.ev_last, 0));
    SubcolIn in{play, cldfrac, alpha};
    if (irng == 0) {
        if (int rc = launch_kiss(s, G.W, ncol, 0, ncol, nlay, icld, permuteseed, in)) return rc;
    } else {
        // one stream over (sub-column, column, layer): drawn chunk-parallel on the device (mt_states), applied per sub-column slab
        HIP_TRY(hipMemsetAsync(G.mask, 0, (size_t)KJ_NWORD * nlay * ncol * sizeof(unsigned), s));
        const int nd = (icld == 4 || icld == 5) ? 2 : 1;
        const unsigned long long per = icld == 3 ? (unsigned long long)ncol : (unsigned long long)ncol * nlay * nd;
        int M = 0;
        unsigned long long C = 0;
        unsigned *states = nullptr;
        if (int rc = mt_states(s, (uint32_t)permuteseed, per, &M, &C, &states)) return rc;
        // as many slabs per launch as fit a 512 MB buffer of deviates (a single slab if it is larger): a small call's 140 slabs are a few launches instead of 140, a large
        // one's launches hold enough chunks (one wavefront each) to fill the device
        const int ns = (int)std::max<unsigned long long>(1ull, std::min<unsigned long long>((unsigned long long)NGPT, (512ull << 20) / (per * 8ull)));     // slabs per pass: at most 512 MB of deviates
        const size_t need = (size_t)ns * per * 8;
        if (G.rnd_bytes < need) {
            if (G.d_rnd) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(G.d_rnd)); G.d_rnd = nullptr; G.rnd_bytes = 0; }
            HIP_TRY(hipMalloc((void **)&G.d_rnd, need));
            G.rnd_bytes = need;
        }
        const dim3 block(BLOCK);
        for (int isub = 0; isub < NGPT; isub += ns) {
            const int n = std::min(ns, NGPT - isub);
            hipLaunchKernelGGL(k_mt_fill, dim3(M, n), dim3(MT_BLOCK), 0, s, (const unsigned *)states, G.d_rnd, isub * M, M, C, per);
            hipLaunchKernelGGL(k_subcol_slab, dim3((ncol + BLOCK - 1) / BLOCK, (n + SLAB_GROUP - 1) / SLAB_GROUP), block, 0, s, G.W, in, (const double *)G.d_rnd, ncol, nlay, icld, isub, n, per);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RRTMG_LW_HIP_EHIP, "generator launch failed: %s", hipGetErrorString(e));
    return 0;
}

int check_subcol_args(int ncol, int nlay, int icld, int *irng)
{
    if (int rc = check_common(ncol, nlay)) return rc;
    if (!irng) return fail(RRTMG_LW_HIP_EARG, "irng is null");
    if (icld < 0 || icld > 5) return fail(RRTMG_LW_HIP_EPHYSICS, "MCICA_SUBCOL: INVALID ICLD");   // src/mcica_subcol_gen_lw.f90:266
    if (*irng != 0) *irng = 1;                                                                    // :442
    return 0;
}

}  // namespace

extern "C" {

const char *rrtmg_lw_hip_last_error(void)
{
    if (tl_err.empty()) { std::lock_guard<std::mutex> lk(g_lasterr_mu); tl_err = g_lasterr; }
    return tl_err.c_str();
}

static int init_state(const char *static_tables_path, const char *kdata_path, double cpdair, int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(RRTMG_LW_HIP_ENODEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(RRTMG_LW_HIP_EARG, "device %d out of range (0..%d)", device, ndev - 1);
    if (!static_tables_path || !kdata_path) return fail(RRTMG_LW_HIP_EARG, "null table path");
    std::string err;
    if (!build_tables(static_tables_path, kdata_path, cpdair, G.H, err)) return fail(RRTMG_LW_HIP_EDATA, "%s", err.c_str());
    HIP_TRY(hipSetDevice(device));
    G.device = device;                  // (from here on the state owns resources on this device: finalize_state releases them there)
    G.sweep_attrs = false;
    if (G.d_ktab) { (void)hipFree(G.d_ktab); G.d_ktab = nullptr; }
    if (G.d_stat) { (void)hipFree(G.d_stat); G.d_stat = nullptr; }
    HIP_TRY(hipMalloc((void **)&G.d_ktab, G.H.ktab.size() * 8));
    HIP_TRY(hipMalloc((void **)&G.d_stat, G.H.stat.size() * 8));
    if (int rc = bounce_h2d(G.d_ktab, G.H.ktab.data(), G.H.ktab.size() * 8)) return rc;          // (never a pageable pointer to the runtime: bounce_h2d)
    if (int rc = bounce_h2d(G.d_stat, G.H.stat.data(), G.H.stat.size() * 8)) return rc;
    if (!G.d_err) {
        HIP_TRY(hipMalloc((void **)&G.d_err, sizeof(int)));
    }
    HIP_TRY(hipMemset(G.d_err, 0, sizeof(int)));
    if (!G.stream) HIP_TRY(hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking));
    HIP_TRY(hipDeviceGetAttribute(&G.cu_total, hipDeviceAttributeMultiprocessorCount, device));
    DevTables &D = G.D;
    D.ktab = G.d_ktab;
    D.stat = G.d_stat;
    for (int b = 0; b < NBND; b++) {
        D.band[b] = G.H.band[b];
        D.delwave[b] = G.H.delwave[b];
        for (int k = 0; k < 6; k++) D.refrat[b][k] = G.H.refrat[b][k];
    }
    D.sl = G.H.sl;
    D.absice0[0] = G.H.absice0[0]; D.absice0[1] = G.H.absice0[1];
    D.abscld1 = G.H.abscld1; D.absliq0 = G.H.absliq0;
    D.heatfac = G.H.heatfac; D.fluxfac = G.H.fluxfac; D.oneminus = G.H.oneminus; D.bpade = G.H.bpade;
    if (G.ws_base) { (void)hipDeviceSynchronize(); (void)hipFree(G.ws_base); G.ws_base = nullptr; G.ws_nlay = 0; G.ws_ncolb = 0; G.ws_cloud = false; G.ws_mc = false; G.ws_groups = 0; G.ws_slabcols = 0; G.ws_gdp = G.ws_efcl = G.ws_ovl = false; }
    G.device = device;
    G.n1 = false;
    G.init = true;
    G.err.clear();
    tl_err.clear();
    return 0;
}

static void finalize_state();

// What this library was built with (kernels.hip, "Tuning switches"): bit 0 tuning build (the benchmark's kernels only), 1 knock-out (wrong
// results), 2 numerics variant, 3 the 256-g-point configuration, 4 kernel-geometry switches.  The shipped libraries: 0 and 8.
unsigned rrtmg_lw_hip_build_flags(void) { return BUILD_FLAGS; }

int rrtmg_lw_hip_init(const char *static_tables_path, const char *kdata_path, double cpdair, int device)
{
    return rrtmg_lw_hip_init_devices(static_tables_path, kdata_path, cpdair, 1, &device);
}

// Several GPUs from one process: state d drives devices[d].  The host-pointer entries rrtmg_lw_hip_run_nomcica / _run_mcica split
// their columns into ndev contiguous blocks, one host thread and one device (its own PCIe link, copy streams, workspace) per block; the
// device-pointer entries, the queue and the sub-column generator stay on devices[0].
int rrtmg_lw_hip_init_devices(const char *static_tables_path, const char *kdata_path, double cpdair, int ndev, const int *devices)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_cur = &g_states[0];
    if (BUILD_FLAGS & (RRLW_BF_KNOCKOUT | RRLW_BF_NUMERICS)) {      // a measurement build whose results are not the product's
        const char *e = getenv("RRTMG_LW_ALLOW_TUNE_BUILD");
        if (!e || atoi(e) == 0)
            return fail(RRTMG_LW_HIP_EARG, "this library is a measurement build (build flags 0x%x: knock-out or numerics variant); set RRTMG_LW_ALLOW_TUNE_BUILD=1 to use it",
                        BUILD_FLAGS);
    }
    if (ndev < 1 || ndev > MAXDEV || !devices) return fail(RRTMG_LW_HIP_EARG, "ndev must be 1..%d", MAXDEV);
    int rc = 0, done = 0;
    std::string err;
    for (int d = 0; d < ndev && rc == 0; d++) {
        g_cur = &g_states[d];
        rc = init_state(static_tables_path, kdata_path, cpdair, devices[d]);
        if (rc != 0) err = G.err; else done = d + 1;
    }
    if (rc == 0) {
        for (int d = ndev; d < g_ndev; d++) { g_cur = &g_states[d]; finalize_state(); }      // the surplus states of an earlier, larger set: dropped once the new set stands
        g_ndev = ndev;
    } else {
        // nothing half-initialised stays behind: every state this call touched is released (the failed one may hold tables), as are the
        // states of the earlier set - a failed init leaves the library uninitialised, and says so
        for (int d = std::max(g_ndev, done + 1) - 1; d >= 0; d--) { g_cur = &g_states[d]; G.init = G.init || G.d_ktab || G.d_stat || G.d_err || G.stream; finalize_state(); }
        g_ndev = 1;
        g_states[0].err = err;
        tl_err = err;
    }
    g_cur = &g_states[0];
    (void)hipSetDevice(g_states[0].device >= 0 ? g_states[0].device : 0);
    return rc;
}

int rrtmg_lw_hip_num_devices(void) { return g_states[0].init ? g_ndev : 0; }

int rrtmg_lw_hip_kdata_is_standin(void) { return G.init ? (G.H.standin ? 1 : 0) : -1; }

void rrtmg_lw_hip_finalize(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (int d = g_ndev - 1; d >= 0; d--) { g_cur = &g_states[d]; finalize_state(); }
    g_cur = &g_states[0];
    g_ndev = 1;
}

static void finalize_state()
{
    if (!G.init) return;
    const bool first = g_cur == &g_states[0];
    (void)hipSetDevice(G.device);
    (void)hipDeviceSynchronize();
    graphs_clear();
    if (G.cap) { (void)hipStreamDestroy(G.cap); G.cap = nullptr; }
    if (G.ws_base) (void)hipFree(G.ws_base);
    if (G.stage_base) (void)hipFree(G.stage_base);
    if (first) {            // the queue and the generator's caches live on the first device
        if (Q.pinned) { forget_pinned(Q.pinned); (void)hipHostFree(Q.pinned); Q.pinned = nullptr; Q.pinned_doubles = 0; }
        Q.chunks.clear(); Q.ncol = 0; Q.open = false;
        { std::lock_guard<std::mutex> lk(g_scan_mu); g_scan.clear(); }
        g_static.clear();
        for (MtStates &S : g_mt) {
            if (S.dev) (void)hipFree(S.dev);
            if (S.polys) (void)hipFree(S.polys);
            S = MtStates{};
        }
    }
    if (G.mask) (void)hipFree(G.mask);
    if (G.kiss.dev) (void)hipFree(G.kiss.dev);
    if (G.d_rnd) { (void)hipFree(G.d_rnd); G.d_rnd = nullptr; G.rnd_bytes = 0; }
    if (G.d_ktab) (void)hipFree(G.d_ktab);
    if (G.d_stat) (void)hipFree(G.d_stat);
    if (G.d_err) (void)hipFree(G.d_err);
    if (G.stream) (void)hipStreamDestroy(G.stream);
    if (G.cp_in) {
        (void)hipStreamDestroy(G.cp_in);
        (void)hipStreamDestroy(G.cp_out);
        for (int k = 0; k < HOST_SETS; k++) { (void)hipEventDestroy(G.ev_h2d[k]); (void)hipEventDestroy(G.ev_cmp[k]); (void)hipEventDestroy(G.ev_d2h[k]); }
    }
    if (G.aux) {
        (void)hipStreamDestroy(G.aux);
        (void)hipEventDestroy(G.ev_in);
        (void)hipEventDestroy(G.ev_last);
        (void)hipEventDestroy(G.ev_fork);
        (void)hipEventDestroy(G.ev_join);
        for (int k = 0; k < 2; k++) { (void)hipEventDestroy(G.ev_ready[k]); (void)hipEventDestroy(G.ev_layer[k]); (void)hipEventDestroy(G.ev_done[k]); }
    }
    drop_sweep_set(0);
    drop_sweep_set(1);
    if (G.lay_u) (void)hipStreamDestroy(G.lay_u);
    if (G.lay_m) (void)hipStreamDestroy(G.lay_m);
    if (G.h_tot) (void)hipHostFree(G.h_tot);
    for (auto &hs : G.hset) {
        if (hs.in) (void)hipHostFree(hs.in);
        if (hs.out) (void)hipHostFree(hs.out);
        if (hs.fill) (void)hipHostFree(hs.fill);
    }
    G = State();
}

int rrtmg_lw_hip_set_batch(int ncol_batch)
{
    if (ncol_batch == 0) {              // back to the default: by the call's layers (auto_batch)
        for (int d = 0; d < MAXDEV; d++) { g_states[d].batch = DEFAULT_BATCH; g_states[d].batch_set = false; }
        return 0;
    }
    if (ncol_batch < 64) return fail(RRTMG_LW_HIP_EARG, "batch must be >= 64 columns");
    if (ncol_batch > 64 * SORT_MAXBLK) return fail(RRTMG_LW_HIP_EARG, "batch must be <= %d columns", 64 * SORT_MAXBLK);      // k_blocksort orders a batch's 64-column blocks in LDS
    for (int d = 0; d < MAXDEV; d++) { g_states[d].batch = ncol_batch; g_states[d].batch_set = true; }
    return 0;
}

#ifdef RRLW_DBG_DUMP
#ifndef RRLW_TUNE
#error "RRLW_DBG_DUMP belongs to tuning builds (-DRRLW_TUNE)"
#endif
// debugging build only (tools/dbg_codes.py): the cell codes / binary-key words k_layer left in the scratch set 0 (which: 0 gas codes, 1 total codes, 2 fw); info = {ncolb, nlay}
int rrtmg_lw_hip_debug_scratch(int which, void *dst, size_t bytes, int *info)
{
    ENTRY_LOCK;
    HIP_TRY(hipDeviceSynchronize());
    info[0] = G.ws_ncolb; info[1] = G.ws_nlay;
    const void *src = which == 2 ? (const void *)G.scrset[0].fw : (const void *)G.scrset[0].scr[which == 1 ? S_CODET : S_CODE];
    const size_t have = which == 2 ? (size_t)NFW * G.ws_nlay * G.ws_ncolb * 4 : (size_t)NQUAD * G.ws_nlay * G.ws_ncolb * CODE_BYTES;
    HIP_TRY(hipMemcpy(dst, src, std::min(bytes, have), hipMemcpyDeviceToHost));
    return 0;
}
#endif

#ifdef RRLW_LAYER_STAMPS
// diagnostic build only: the cycle sums of k_layer's segments (kernels.hip, STAMP) since the last call; out[NSTAMP] = waves counted
int rrtmg_lw_hip_debug_stamps(unsigned long long *out, int n)
{
    ENTRY_LOCK;
    unsigned long long h[NSTAMP + 1] = {};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof h));
    for (int i = 0; i < n && i <= NSTAMP; i++) out[i] = h[i];
    unsigned long long z[NSTAMP + 1] = {};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z));
    return 0;
}
#endif

int rrtmg_lw_hip_set_overlap(int on)
{
    ENTRY_LOCK;
    for (int d = 0; d < MAXDEV; d++) g_states[d].split_sweep = on != 0;       // the second scratch set is allocated by the next call that needs it
    return 0;
}

// Cloudy batches of up to `ncol` columns take ONE sweep launch per band group (the cloud-zone kernel over all levels) instead of three
// (one_sweep, above); 0 = never.  Results do not depend on it.  Returns the previous value.
int rrtmg_lw_hip_set_one_sweep_max(int ncol)
{
    ENTRY_LOCK;
    const int prev = g_one_sweep_max;
    g_one_sweep_max = ncol < 0 ? 0 : ncol;
    return prev;
}

// Device-resident one-batch calls of up to `ncol` columns are replayed as one graph from their third occurrence on (run_pipelined);
// 0 = never.  Results do not depend on it.  Returns the previous value.
int rrtmg_lw_hip_set_graph_max(int ncol)
{
    ENTRY_LOCK;
    const int prev = g_graph_max;
    g_graph_max = ncol < 0 ? 0 : ncol;
    return prev;
}
// graphs captured / calls replayed from a graph since the library was initialised (first device)
void rrtmg_lw_hip_graph_stats(long long *captures, long long *replays)
{
    ENTRY_LOCK;
    if (captures) *captures = g_states[0].graph_captures;
    if (replays) *replays = g_states[0].graph_replays;
}

// Batches of up to `ncol` columns are swept one band per workgroup (g_split_max); 0 = never.  Results do not depend on it.  Returns the
// previous value.
int rrtmg_lw_hip_set_split_max(int ncol)
{
    ENTRY_LOCK;
    const int prev = g_split_max;
    g_split_max = ncol < 0 ? 0 : ncol;
    return prev;
}

// k_layer's second pass with the wide staging window for (window, layer) pairs whose columns lie more than one reference-pressure plane
// apart (terrain-following grids): on = 1 (default) / off = 0.  Results do not depend on it.  Returns the previous value.
int rrtmg_lw_hip_set_wide_window(int on)
{
    ENTRY_LOCK;
    const int prev = g_wide_window ? 1 : 0;
    g_wide_window = on != 0;
    return prev;
}

// k_layer's sixteen bands of a (window of 256 columns, layer) over two or four workgroups where the batch has too few such pairs to fill the
// chip (up to ~2 000 columns of 72 layers): on = 1 (default) / off = 0.  Results do not depend on it.  Returns the previous value.
int rrtmg_lw_hip_set_layer_split(int on)
{
    ENTRY_LOCK;
    const int prev = g_layer_split ? 1 : 0;
    g_layer_split = on != 0;
    return prev;
}

// The columns of a cloudy non-McICA batch are taken in k_colsort's order (by cloud top within windows of 256 columns): on = 1 / off = 0.
// min_gain >= 0: the block-levels a window's reordering must take out of the cloud zone (k_colsort; < 0 keeps the value).
// Results do not depend on it.  Returns the previous value.
int rrtmg_lw_hip_set_column_sort(int on, int min_gain)
{
    ENTRY_LOCK;
    const int prev = g_colsort ? 1 : 0;
    g_colsort = on != 0;
    if (min_gain >= 0) g_colsort_min = std::min(min_gain, COLSORT_NEVER);      // (k_colsort compares gain x 4 with min_gain x 4 in 32 bits)
    return prev;
}

// the threshold in force (rrtmg_lw_hip_set_column_sort's min_gain): what a caller that changes it for a while puts back
int rrtmg_lw_hip_column_sort_min(void)
{
    ENTRY_LOCK;
    return g_colsort_min;
}

// CU partition of the overlapped pipeline (device-pointer entries): k_layer of batch i + 1 on `layer_cus` CUs (rounded to a multiple of 8:
// the same number from every XCD), the sweeps and k_flux of batch i on the others; 0 = no partition.  A value > 0 switches the overlap on.
int rrtmg_lw_hip_set_cu_partition(int layer_cus)
{
    ENTRY_LOCK;
    if (!G.init) return fail(RRTMG_LW_HIP_ENOTINIT, "rrtmg_lw_hip_init has not been called");
    State *const keep = g_cur;
    int rc = 0;
    for (int d = 0; d < g_ndev && rc == 0; d++) {
        g_cur = &g_states[d];
        (void)hipSetDevice(G.device);
        (void)hipDeviceSynchronize();
        int n = layer_cus <= 0 ? 0 : std::max(8, std::min(G.cu_total - 8, (layer_cus + 4) / 8 * 8));
        if (G.cu_total < 16) n = 0;
        drop_sweep_set(1);
        if (G.lay_u) { (void)hipStreamDestroy(G.lay_u); G.lay_u = nullptr; }
        if (G.lay_m) { (void)hipStreamDestroy(G.lay_m); G.lay_m = nullptr; }
        G.cu_layer = n;
        if (n > 0) {
            G.split_sweep = true;
            // (created here so that a refusal of CU masks by the runtime is reported by this call)
            if ((rc = make_stream(&G.lay_u, 0, 0)) == 0 && (rc = make_stream(&G.lay_m, 0, n)) == 0) rc = ensure_sweep_set(1);
            if (rc != 0 && d > 0) g_states[0].err = G.err;
        }
    }
    g_cur = keep;
    (void)hipSetDevice(G.device);
    return rc;
}
int rrtmg_lw_hip_cu_partition(void) { return g_states[0].cu_layer; }

// measurement only (tools/n1_expf_measure.sh, the prototype's GPU test): cloud-free GCM calls take k_n1 instead of the production sweeps.
// An explicit call, not an environment variable: a stray variable in a job's environment must not change the numerical path.
int rrtmg_lw_hip_set_n1_prototype(int on)
{
    ENTRY_LOCK;
    if (on && CODE_BITS != 32) return fail(RRTMG_LW_HIP_EARG, "the k_n1 prototype reads 32-bit cell codes");
    G.n1 = on != 0;
    return 0;
}
int rrtmg_lw_hip_n1_prototype(void) { return g_states[0].n1 ? 1 : 0; }

// device memory the library holds right now, over all its devices: workspaces, host-entry staging, sub-column masks, the slab buffer and
// the cached chunk states of the Mersenne-Twister stream, the kissvec jump table
long long rrtmg_lw_hip_workspace_bytes(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    size_t tot = 0;
    for (int d = 0; d < g_ndev; d++) {
        const State &S = g_states[d];
        tot += S.ws_bytes + S.stage_bytes + S.mask_bytes + S.rnd_bytes + (S.kiss.dev ? sizeof(KissJump) * (KJ_NGROUP + 1) : 0);
    }
    for (const MtStates &M : g_mt) tot += M.bytes();
    return (long long)tot;
}
int rrtmg_lw_hip_num_chunks(void) { return NQUAD; }
int rrtmg_lw_hip_gpoints(void) { return NGPT; }

void rrtmg_lw_hip_profile_begin(void)
{
    ENTRY_LOCK;
    for (auto &r : G.prof) { G.evpool.push_back(r.a); G.evpool.push_back(r.b); }
    G.prof.clear();
    G.profile = true;
}

// Stops event collection, synchronises the device and writes one line per kernel name:
// "<name> <launches> <total_ms>\n".  Returns the number of bytes written (truncated to len-1).
int rrtmg_lw_hip_profile_end(char *buf, int len)
{
    ENTRY_LOCK;
    G.profile = false;
    (void)hipDeviceSynchronize();
    std::vector<std::string> names;
    std::vector<double> total;
    std::vector<int> count;
    for (auto &r : G.prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = 0.f;
        size_t k = 0;
        for (; k < names.size(); k++) if (names[k] == r.name) break;
        if (k == names.size()) { names.push_back(r.name); total.push_back(0.0); count.push_back(0); }
        total[k] += ms;
        count[k] += 1;
        G.evpool.push_back(r.a);
        G.evpool.push_back(r.b);
    }
    G.prof.clear();
    std::string out;
    for (size_t k = 0; k < names.size(); k++) {
        char line[160];
        snprintf(line, sizeof line, "%s %d %.6f\n", names[k].c_str(), count[k], total[k]);
        out += line;
    }
    if (buf && len > 0) {
        const size_t n = std::min(out.size(), (size_t)len - 1);
        memcpy(buf, out.data(), n);
        buf[n] = 0;
        return (int)n;
    }
    return 0;
}

int rrtmg_lw_hip_check(void *stream)
{
    ENTRY_LOCK;
    if (!G.init) return fail(RRTMG_LW_HIP_ENOTINIT, "rrtmg_lw_hip_init has not been called");
    if (g_ndev <= 1) return read_physics_error((hipStream_t)stream);
    // several states (rrtmg_lw_hip_init_devices): the device-pointer entries run on the state of the arrays' device - wait for the stream,
    // then look at every state's error word (the first error found is reported)
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    State *const keep = g_cur;
    int rc = 0;
    for (int d = 0; d < g_ndev && rc == 0; d++) {
        g_cur = &g_states[d];
        if (!G.init) continue;
        DeviceGuard dg2;
        rc = read_physics_error(nullptr);
        if (rc != 0 && d > 0) { const std::string e = G.err; g_cur = keep; G.err = e; }
    }
    g_cur = keep;
    return rc;
}
// the state (index into the devices of rrtmg_lw_hip_init_devices) this thread's last device-pointer entry ran on
int rrtmg_lw_hip_last_device_state(void) { return tl_dev_state; }

int rrtmg_lw_hip_run_nomcica_device(
    int ncol, int nlay, int *icld, int idrv,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt, void *stream)
{
    ENTRY_LOCK_FOR(play);
    if (int rc = check_common(ncol, nlay)) return rc;
    if (!icld) return fail(RRTMG_LW_HIP_EARG, "icld is null");
    if (*icld < 0 || *icld > 3) *icld = 2;                       // src/rrtmg_lw_rad.nomcica.f90:456
    if (idrv == 1 && (!duflx_dt || !duflxc_dt)) return fail(RRTMG_LW_HIP_EARG, "idrv=1 needs duflx_dt and duflxc_dt");
    const int mode = *icld == 0 ? 0 : (*icld == 1 ? 1 : 2);      // :546-560 (icld=0 -> rtrnmr clear branch)
    const int nbmax = balanced_batch(ncol, eff_batch(nlay));
    if (int rc = ensure_workspace(nlay, nbmax, mode != 0, false, idrv, mode)) return rc;
    GcmIn g{play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr,
            ccl4vmr, emis, cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer};
    FluxOut out{uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt, nullptr, nullptr};
    return run_pipelined((hipStream_t)stream, ncol, nlay, mode, idrv, g, inflglw, iceflglw, liqflglw, out, nullptr);
}

}   // extern "C"

namespace {
// columns [c0, c1) of the caller's host arrays (ncol columns wide) on the calling thread's current device state; no lock taken
int nomcica_host_range(int ncol, int c0, int c1, int nlay, int icld, int idrv,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt)
{
    if (int rc = check_common(ncol, nlay)) return rc;
    HIP_TRY(hipDeviceSynchronize());        // asynchronous device-entry work of earlier calls shares the workspace
    const int mode = icld == 0 ? 0 : (icld == 1 ? 1 : 2);
    const bool cloud = icld >= 1;           // inatm copies the cloud arrays only when icld >= 1 (:893-910)
    // host arrays: the copies bound the rate (PCIe), and they overlap with the kernels only across batches - smaller batches than
    // the device-resident default
    const int nbmax = balanced_batch(c1 - c0, std::min(eff_batch(nlay), HOST_BATCH));
    if (int rc = ensure_workspace(nlay, nbmax, mode != 0, false, idrv, mode)) return rc;
    const size_t L = (size_t)nlay;
    // What the copies need not carry (62 % of a column's bytes are taucld and tauaer, 2 x 16 nlay values):
    //  * with inflglw >= 1 cldprop reads taucld only through the sum over the bands, tauctot (src/rrtmg_lw_cldprop.f90:173-186): the sum is
    //    formed here, in the reference's order, on the host threads - one value per (column, layer) travels instead of sixteen;
    //  * rows that hold one value for all columns of a batch - zero aerosol / cloud rows, well-mixed gases - are not copied (stage_rows).
    const bool use_tot = cloud && inflglw != 0;
    std::vector<HostIn> ins = {
        {play, 1, L, 0}, {plev, 1, L + 1, 0}, {tlay, 1, L, 0}, {tlev, 1, L + 1, 0}, {tsfc, 1, 1, 0},
        {h2ovmr, 1, L, 0}, {o3vmr, 1, L, 0}, {co2vmr, 1, L, 0}, {ch4vmr, 1, L, 0}, {n2ovmr, 1, L, 0}, {o2vmr, 1, L, 0},
        {cfc11vmr, 1, L, 0}, {cfc12vmr, 1, L, 0}, {cfc22vmr, 1, L, 0}, {ccl4vmr, 1, L, 0}, {emis, 1, 16, 0}, {tauaer, 1, 16 * L, 0},
        {cloud ? cldfr : nullptr, 1, L, 0}, {cloud ? taucld : nullptr, (size_t)(use_tot ? 1 : NBND), L, 0}, {cloud ? cicewp : nullptr, 1, L, 0},
        {cloud ? cliqwp : nullptr, 1, L, 0}, {cloud ? reice : nullptr, 1, L, 0}, {cloud ? reliq : nullptr, 1, L, 0}};
    for (size_t k = 0; k < 17; k++) if (!ins[k].h) return fail(RRTMG_LW_HIP_EARG, "null input array (argument %d)", (int)k);
    if (cloud) for (size_t k = 17; k < ins.size(); k++) if (!ins[k].h) return fail(RRTMG_LW_HIP_EARG, "null cloud array");
    if (use_tot && G.h_tot_doubles < HOST_SETS * L * (size_t)nbmax) {
        if (G.h_tot) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipHostFree(G.h_tot)); G.h_tot = nullptr; G.h_tot_doubles = 0; }
        HIP_TRY(hipHostMalloc((void **)&G.h_tot, HOST_SETS * L * (size_t)nbmax * sizeof(double), hipHostMallocDefault));
        G.h_tot_doubles = HOST_SETS * L * (size_t)nbmax;
    }
    std::vector<unsigned char> cloudfree(L, 0), cf_uni(L, 0);
    std::vector<uint64_t> cf_bits(L, 0);
    auto prep = [&](int k, int col0, int nb, hipStream_t) -> int {
        if (!cloud) return 0;
        // layers without cloud in any column of the batch: the other five cloud arrays are not read there (nor summed, nor scanned, nor copied)
        rows_below(cldfr, 1, L, (size_t)ncol, (size_t)col0, (size_t)nb, 1.e-20, cloudfree.data(), cf_uni.data(), cf_bits.data());
        ins[17].known = cf_uni.data(); ins[17].known_bits = cf_bits.data();        // (those layers of cldfr have just been read to their end)
        for (size_t a = 18; a < ins.size(); a++) ins[a].skip = cloudfree.data();
        if (!use_tot) return 0;
        double *tot = G.h_tot + (size_t)k * L * (size_t)nbmax;        // (the pipeline has waited for the copies that read scratch set k last)
        size_t cloudy_layers = 0;
        for (size_t l = 0; l < L; l++) cloudy_layers += cloudfree[l] ? 0 : 1;
        host_parallel((size_t)NBND * cloudy_layers * (size_t)nb * 8, [&](int t, int nt) {
            // (pieces of 4 096 columns of a layer, dealt round-robin: the cloudy layers are few)
            const size_t piece = 4096, per = ((size_t)nb + piece - 1) / piece;
            size_t task = 0;
            for (size_t lay = 0; lay < L; lay++) {
                if (cloudfree[lay]) continue;
                for (size_t pc = 0; pc < per; pc++, task++) {
                    if (task % (size_t)nt != (size_t)t) continue;
                    const size_t c1 = std::min((size_t)nb, (pc + 1) * piece);
                    for (size_t c = pc * piece; c < c1; c++) {
                        const double *p = taucld + (size_t)NBND * ((size_t)col0 + c + (size_t)ncol * lay);
                        double sum = 0.0;
                        for (int ib = 0; ib < NBND; ib++) sum = sum + p[ib];
                        tot[lay * (size_t)nb + c] = sum;
                    }
                }
            }
        });
        ins[18].src = tot; ins[18].src_ncol = (size_t)nb; ins[18].src_col0 = 0; ins[18].src_pinned = true;      // rows of nb sums, from the pinned scratch
        return 0;
    };
    std::vector<HostOut> outs = {{uflx, L + 1, 0, true}, {dflx, L + 1, 0, true}, {hr, L, 0, true}, {uflxc, L + 1, 0, true},
                                 {dflxc, L + 1, 0, true}, {hrc, L, 0, true}, {duflx_dt, L + 1, 0, idrv == 1}, {duflxc_dt, L + 1, 0, idrv == 1}};
    for (size_t k = 0; k < 6; k++) if (!outs[k].h) return fail(RRTMG_LW_HIP_EARG, "null output array");
    auto body = [&](hipStream_t s, int nb, int, std::vector<HostIn> &in, std::vector<HostOut> &out_) -> int {
        GcmIn g{in[0].d, in[1].d, in[2].d, in[3].d, in[4].d, in[5].d, in[6].d, in[7].d, in[8].d, in[9].d, in[10].d,
                in[11].d, in[12].d, in[13].d, in[14].d, in[15].d, in[17].d, use_tot ? nullptr : in[18].d, in[19].d, in[20].d, in[21].d, in[22].d, in[16].d,
                use_tot ? in[18].d : nullptr};
        ColIn c{};
        FluxOut out{out_[0].d, out_[1].d, out_[2].d, out_[3].d, out_[4].d, out_[5].d, out_[6].d, out_[7].d, nullptr, nullptr};
        return run_batch<true>(s, nb, 0, nb, nlay, mode, idrv, 1, 16, g, c, inflglw, iceflglw, liqflglw, out);
    };
    if (int rc = host_pipeline(ncol, c0, c1, nbmax, ins, outs, body, prep)) return rc;
    hipStream_t s = G.stream;
    return read_physics_error(s);
}

// The columns of a host-pointer call over the devices of rrtmg_lw_hip_init_devices: contiguous blocks (multiples of 64 columns), one host
// thread per device; every block runs the one-device pipeline on its own state (H2D | kernels | D2H on that device's streams).  Columns
// are independent (src/rrtmg_lw_rad.nomcica.f90:472), so the results do not depend on the split.  Caller holds the entry lock.
template <class RangeFn>
int fan_out(int ncol, RangeFn range)
{
    State *const home = g_cur;
    if (g_ndev <= 1 || ncol < 128) return range(0, ncol);
    const int per = (int)align_up((size_t)(ncol + g_ndev - 1) / g_ndev, 64);
    int rcs[MAXDEV] = {};
    std::vector<std::thread> th;
    for (int d = 0; d < g_ndev; d++) {
        const int c0 = d * per, c1 = std::min(ncol, c0 + per);
        if (c0 >= c1) break;
        th.emplace_back([&rcs, &range, d, c0, c1]() {
            g_cur = &g_states[d];
            tl_pool = d; tl_share = g_ndev;
            if (hipSetDevice(G.device) != hipSuccess) { rcs[d] = fail(RRTMG_LW_HIP_EHIP, "hipSetDevice(%d) failed", G.device); return; }
            rcs[d] = range(c0, c1);
        });
    }
    for (auto &x : th) x.join();
    for (int d = 0; d < g_ndev; d++)
        if (rcs[d] != 0) { if (&g_states[d] != home) home->err = g_states[d].err; tl_err = g_states[d].err; return rcs[d]; }       // (the worker's thread-local text is not this thread's)
    return 0;
}

int nomcica_host(int ncol, int nlay, int *icld, int idrv,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt)
{
    if (int rc = check_common(ncol, nlay)) return rc;
    if (!icld) return fail(RRTMG_LW_HIP_EARG, "icld is null");
    if (*icld < 0 || *icld > 3) *icld = 2;
    if (idrv == 1 && (!duflx_dt || !duflxc_dt)) return fail(RRTMG_LW_HIP_EARG, "idrv=1 needs duflx_dt and duflxc_dt");
    const int ic = *icld;
    return fan_out(ncol, [&](int c0, int c1) {
        return nomcica_host_range(ncol, c0, c1, nlay, ic, idrv, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis,
                              inflglw, iceflglw, liqflglw, cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt);
    });
}
bool comb_enabled();
int comb_max();
long long comb_calls_total();
long long comb_passes_total();
int nomcica_combined(int ncol, int nlay, int *icld, int idrv,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt);
}   // namespace

extern "C" {

int rrtmg_lw_hip_run_nomcica(
    int ncol, int nlay, int *icld, int idrv,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt)
{
    if (ncol >= 1 && ncol <= comb_max() && icld && comb_enabled()) return nomcica_combined(ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis,
                              inflglw, iceflglw, liqflglw, cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt);
    ENTRY_LOCK;
    return nomcica_host(ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis,
                              inflglw, iceflglw, liqflglw, cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt);
}

// calls and device passes of the combining entry since the library was loaded (a pass serves one or more calls)
void rrtmg_lw_hip_combine_stats(long long *calls, long long *passes)
{
    if (calls) *calls = comb_calls_total();
    if (passes) *passes = comb_passes_total();
}

int rrtmg_lw_hip_run_columns(
    int ncol, int nlayers, int istart, int iend, int icld, int idrv,
    const double *pavel, const double *tavel, const double *pz, const double *tz, const double *tbound,
    const double *semiss, const double *coldry, const double *wkl, const double *wbrodl, const double *wx,
    const double *pwvcm, int inflag, int iceflag, int liqflag, const double *cldfrac, const double *tauc,
    const double *ciwp, const double *clwp, const double *rei, const double *rel, const double *taua,
    double *totuflux, double *totdflux, double *fnet, double *htr,
    double *totuclfl, double *totdclfl, double *fnetc, double *htrc,
    double *dtotuflux_dt, double *dtotuclfl_dt)
{
    ENTRY_LOCK;
    if (int rc = check_common(ncol, nlayers)) return rc;
    if (G.init) HIP_TRY(hipDeviceSynchronize());      // asynchronous device-entry work of earlier calls shares the workspace
    if (istart < 1 || iend > 16 || istart > iend) return fail(RRTMG_LW_HIP_EARG, "bad band range %d..%d", istart, iend);
    if (ncol > eff_batch(nlayers)) return fail(RRTMG_LW_HIP_EARG, "run_columns handles at most one batch (%d columns)", eff_batch(nlayers));
    const int mode = icld == 0 ? 0 : (icld == 1 ? 1 : 2);
    if (int rc = ensure_workspace(nlayers, ncol, true, false, idrv, mode)) return rc;
    const size_t n = (size_t)ncol, L = (size_t)nlayers;
    struct In { const double *h; size_t cnt; double *d; };
    In ins[] = {{pavel, n * L, 0}, {tavel, n * L, 0}, {pz, n * (L + 1), 0}, {tz, n * (L + 1), 0}, {tbound, n, 0}, {semiss, 16 * n, 0},
                {coldry, n * L, 0}, {wkl, 7 * n * L, 0}, {wbrodl, n * L, 0}, {wx, 4 * n * L, 0}, {pwvcm, n, 0}, {cldfrac, n * L, 0},
                {tauc, 16 * n * L, 0}, {ciwp, n * L, 0}, {clwp, n * L, 0}, {rei, n * L, 0}, {rel, n * L, 0}, {taua, 16 * n * L, 0}};
    size_t tot = 0;
    for (auto &i : ins) tot += i.cnt;
    const size_t out_d = 10 * n * (L + 1);
    if (int rc = ensure_stage((tot + out_d) * 8 + 4096)) return rc;
    double *p = (double *)G.stage_base;
    hipStream_t s = G.stream;
    for (auto &i : ins) {
        i.d = p; p += i.cnt;
        if (int rc = bounce_h2d(i.d, i.h, i.cnt * 8)) return rc;
    }
    double *o[10];
    for (auto &q : o) { q = p; p += n * (L + 1); }
    ColIn c{ins[0].d, ins[1].d, ins[2].d, ins[3].d, ins[4].d, ins[5].d, ins[6].d, ins[7].d, ins[8].d, ins[9].d, ins[10].d,
            ins[11].d, ins[12].d, ins[13].d, ins[14].d, ins[15].d, ins[16].d, ins[17].d};
    GcmIn g{};
    // o: 0 uflx 1 dflx 2 fnet 3 htr 4 uclfl 5 dclfl 6 fnetc 7 htrc 8 duflx 9 duclfl ; htr arrays hold (ncol, 0:nlayers), top level = 0
    HIP_TRY(hipMemsetAsync(o[3], 0, n * (L + 1) * 8, s));
    HIP_TRY(hipMemsetAsync(o[7], 0, n * (L + 1) * 8, s));
    HIP_TRY(hipMemsetAsync(o[8], 0, n * (L + 1) * 8, s));
    HIP_TRY(hipMemsetAsync(o[9], 0, n * (L + 1) * 8, s));
    FluxOut out{o[0], o[1], o[3], o[4], o[5], o[7], o[8], o[9], o[2], o[6]};
    if (int rc = run_batch<false>(s, ncol, 0, ncol, nlayers, mode, idrv, istart, iend, g, c, inflag, iceflag, liqflag, out)) return rc;
    double *ho[10] = {totuflux, totdflux, fnet, htr, totuclfl, totdclfl, fnetc, htrc, dtotuflux_dt, dtotuclfl_dt};
    HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < 10; k++) if (int rc = bounce_d2h(ho[k], o[k], n * (L + 1) * 8)) return rc;
    return read_physics_error(s);
}

// McICA flavour of the prepared-column entry: cldprmc -> setcoef -> taumol -> rtrnmc for `ncol` (column, sample) pairs.
int rrtmg_lw_hip_run_columns_mcica(
    int ncol, int nlayers, int istart, int iend, int icld, int idrv,
    const double *pavel, const double *tavel, const double *pz, const double *tz, const double *tbound,
    const double *semiss, const double *coldry, const double *wkl, const double *wbrodl, const double *wx,
    const double *pwvcm, int inflag, int iceflag, int liqflag, const double *cldfmc, const double *taucmc,
    const double *ciwpmc, const double *clwpmc, const double *reicmc, const double *relqmc, const double *taua,
    double *totuflux, double *totdflux, double *fnet, double *htr,
    double *totuclfl, double *totdclfl, double *fnetc, double *htrc,
    double *dtotuflux_dt, double *dtotuclfl_dt)
{
    ENTRY_LOCK;
    if (int rc = check_mcica_build()) return rc;
    if (int rc = check_common(ncol, nlayers)) return rc;
    if (G.init) HIP_TRY(hipDeviceSynchronize());      // asynchronous device-entry work of earlier calls shares the workspace
    if (istart < 1 || iend > 16 || istart > iend) return fail(RRTMG_LW_HIP_EARG, "bad band range %d..%d", istart, iend);
    if (ncol > eff_batch(nlayers)) return fail(RRTMG_LW_HIP_EARG, "run_columns_mcica handles at most one batch (%d columns)", eff_batch(nlayers));
    const int mode = icld == 0 ? 0 : 3;
    if (int rc = ensure_workspace(nlayers, ncol, true, true, idrv, mode)) return rc;
    const size_t n = (size_t)ncol, L = (size_t)nlayers;
    struct In { const double *h; size_t cnt; double *d; };
    In ins[] = {{pavel, n * L, 0}, {tavel, n * L, 0}, {pz, n * (L + 1), 0}, {tz, n * (L + 1), 0}, {tbound, n, 0}, {semiss, 16 * n, 0},
                {coldry, n * L, 0}, {wkl, 7 * n * L, 0}, {wbrodl, n * L, 0}, {wx, 4 * n * L, 0}, {pwvcm, n, 0},
                {cldfmc, NGPT * n * L, 0}, {taucmc, NGPT * n * L, 0}, {ciwpmc, NGPT * n * L, 0}, {clwpmc, NGPT * n * L, 0},
                {reicmc, n * L, 0}, {relqmc, n * L, 0}, {taua, 16 * n * L, 0}};
    size_t tot = 0;
    for (auto &i : ins) { if (!i.h) return fail(RRTMG_LW_HIP_EARG, "null input array"); tot += i.cnt; }
    const size_t out_d = 10 * n * (L + 1);
    if (int rc = ensure_stage((tot + out_d) * 8 + 4096)) return rc;
    double *p = (double *)G.stage_base;
    hipStream_t s = G.stream;
    for (auto &i : ins) {
        i.d = p; p += i.cnt;
        if (int rc = bounce_h2d(i.d, i.h, i.cnt * 8)) return rc;
    }
    double *o[10];
    for (auto &q : o) { q = p; p += n * (L + 1); }
    ColIn c{ins[0].d, ins[1].d, ins[2].d, ins[3].d, ins[4].d, ins[5].d, ins[6].d, ins[7].d, ins[8].d, ins[9].d, ins[10].d,
            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ins[17].d};
    McIn m{ins[11].d, ins[12].d, ins[13].d, ins[14].d, ins[15].d, ins[16].d};
    GcmIn g{};
    HIP_TRY(hipMemsetAsync(o[3], 0, n * (L + 1) * 8, s));
    HIP_TRY(hipMemsetAsync(o[7], 0, n * (L + 1) * 8, s));
    HIP_TRY(hipMemsetAsync(o[8], 0, n * (L + 1) * 8, s));
    HIP_TRY(hipMemsetAsync(o[9], 0, n * (L + 1) * 8, s));
    FluxOut out{o[0], o[1], o[3], o[4], o[5], o[7], o[8], o[9], o[2], o[6]};
    if (int rc = run_batch<false>(s, ncol, 0, ncol, nlayers, mode, idrv, istart, iend, g, c, inflag, iceflag, liqflag, out, &m)) return rc;
    double *ho[10] = {totuflux, totdflux, fnet, htr, totuclfl, totdclfl, fnetc, htrc, dtotuflux_dt, dtotuclfl_dt};
    HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < 10; k++) if (int rc = bounce_d2h(ho[k], o[k], n * (L + 1) * 8)) return rc;
    return read_physics_error(s);
}

// Pin a host array the caller will pass repeatedly (a GCM's profile and flux arrays live for the whole run): the H2D / D2H copies
// of the host-pointer entries then run as asynchronous DMA at PCIe rate instead of through the runtime's pageable staging.
int rrtmg_lw_hip_host_register(void *ptr, long long bytes)
{
    ENTRY_LOCK;
    if (!G.init) return fail(RRTMG_LW_HIP_ENOTINIT, "rrtmg_lw_hip_init has not been called");
    if (!ptr || bytes <= 0) return fail(RRTMG_LW_HIP_EARG, "bad host range");
    HIP_TRY(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
    g_pinned.emplace_back((const char *)ptr, (size_t)bytes);
    return 0;
}

int rrtmg_lw_hip_host_is_registered(const void *ptr, long long bytes)
{
    ENTRY_LOCK;
    return (ptr && bytes > 0 && host_range_pinned(ptr, (size_t)bytes)) ? 1 : 0;
}

int rrtmg_lw_hip_host_unregister(void *ptr)
{
    ENTRY_LOCK;
    if (!ptr) return fail(RRTMG_LW_HIP_EARG, "null pointer");
    HIP_TRY(hipHostUnregister(ptr));
    forget_pinned(ptr);
    return 0;
}

// The caller declares [ptr, ptr + bytes) static: its contents stay as they are until rrtmg_lw_hip_host_changed(ptr).  The host-pointer entries
// then scan each of its rows once per column batch shape instead of once per call (g_scan).  Nothing else changes: an array that is not
// declared is scanned on every call.
int rrtmg_lw_hip_host_static(const void *ptr, long long bytes)
{
    ENTRY_LOCK;
    if (!ptr || bytes <= 0) return fail(RRTMG_LW_HIP_EARG, "bad range");
    for (auto &r : g_static)
        if (r.p == (const char *)ptr) { r.bytes = (size_t)bytes; r.gen = ++g_static_gen; return 0; }
    g_static.push_back(StaticRange{(const char *)ptr, (size_t)bytes, ++g_static_gen});
    return 0;
}
// The contents of the static range that starts at ptr have changed: its cached row scans are dropped and the next call scans it again.
// keep = 0 also withdraws the declaration (before the array is freed).
int rrtmg_lw_hip_host_changed(const void *ptr, int keep)
{
    ENTRY_LOCK;
    for (size_t i = 0; i < g_static.size(); i++) {
        if (g_static[i].p != (const char *)ptr) continue;
        const char *lo = g_static[i].p, *hi = lo + g_static[i].bytes;
        {
            std::lock_guard<std::mutex> lk(g_scan_mu);
            for (auto it = g_scan.begin(); it != g_scan.end();)
                if ((const char *)it->first.base >= lo && (const char *)it->first.base < hi) it = g_scan.erase(it); else ++it;
        }
        if (keep) g_static[i].gen = ++g_static_gen; else g_static.erase(g_static.begin() + (long)i);
        return 0;
    }
    return fail(RRTMG_LW_HIP_EARG, "no static range starts at this address");
}

// Streams `bytes` from one device buffer into another with 16 B per lane (k_calibrate): known traffic for PMC calibration.
int rrtmg_lw_hip_calibrate_stream(long long bytes)
{
    ENTRY_LOCK;
    if (!G.init) return fail(RRTMG_LW_HIP_ENOTINIT, "rrtmg_lw_hip_init has not been called");
    if (bytes < 16) return fail(RRTMG_LW_HIP_EARG, "bytes must be >= 16");
    const size_t n = (size_t)bytes / 16;
    void *a = nullptr, *b = nullptr;
    HIP_TRY(hipMalloc(&a, n * 16));
    if (hipMalloc(&b, n * 16) != hipSuccess) { (void)hipFree(a); return fail(RRTMG_LW_HIP_EHIP, "hipMalloc failed"); }
    (void)hipMemset(a, 0, n * 16);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_calibrate, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, G.stream, (const double2 *)a, (double2 *)b, n);
    hipError_t e = hipStreamSynchronize(G.stream);
    (void)hipFree(a);
    (void)hipFree(b);
    if (e != hipSuccess) return fail(RRTMG_LW_HIP_EHIP, "calibration kernel failed: %s", hipGetErrorString(e));
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Aggregation of small calls (include/rrtmg_lw_hip.h): chunks are recorded, packed column-wise into one pinned staging set and
// solved in one pass through the host-pointer entry.
// ---------------------------------------------------------------------------------------------------
}   // extern "C"

namespace {
// inner extent and number of rows of the 23 inputs ([rows][ncol][inner], run_nomcica's order with tauaer last) and 8 outputs
void queue_shapes(int nlay, size_t (&in_inner)[23], size_t (&in_rows)[23], size_t (&out_rows)[8])
{
    const size_t L = (size_t)nlay;
    const size_t rows[23] = {L, L + 1, L, L + 1, 1, L, L, L, L, L, L, L, L, L, L, 16, L, L, L, L, L, L, 16 * L};
    for (int k = 0; k < 23; k++) { in_inner[k] = 1; in_rows[k] = rows[k]; }
    in_inner[17] = NBND;                      // taucld (16, ncol, nlay)
    const size_t orows[8] = {L + 1, L + 1, L, L + 1, L + 1, L, L + 1, L + 1};
    for (int k = 0; k < 8; k++) out_rows[k] = orows[k];
}
}   // namespace

extern "C" {

int rrtmg_lw_hip_queue_begin(int nlay, int icld, int idrv, int inflglw, int iceflglw, int liqflglw)
{
    ENTRY_LOCK;
    if (int rc = check_common(1, nlay)) return rc;
    if (Q.open && !Q.chunks.empty()) return fail(RRTMG_LW_HIP_EARG, "%lld queued columns have not been flushed", Q.ncol);
    Q.chunks.clear();
    Q.ncol = 0;
    Q.nlay = nlay; Q.icld = icld; Q.idrv = idrv; Q.inflg = inflglw; Q.iceflg = iceflglw; Q.liqflg = liqflglw;
    Q.open = true;
    return 0;
}

int rrtmg_lw_hip_queue_columns(void) { std::lock_guard<std::mutex> lk(g_mu); return (int)Q.ncol; }

int rrtmg_lw_hip_queue_add(
    int ncol, int *icld,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt)
{
    ENTRY_LOCK;
    if (!Q.open) return fail(RRTMG_LW_HIP_EARG, "rrtmg_lw_hip_queue_begin has not been called");
    if (ncol < 1) return fail(RRTMG_LW_HIP_EARG, "bad chunk size %d", ncol);
    if (icld && *icld != Q.icld) return fail(RRTMG_LW_HIP_EARG, "chunk icld %d differs from the queue's %d", *icld, Q.icld);
    QueuedChunk c{ncol, icld,
                  {play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis,
                   cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer},
                  {uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt}};
    const int icld_eff = (Q.icld < 0 || Q.icld > 3) ? 2 : Q.icld;
    for (int k = 0; k < 23; k++) {
        const bool cloud_arr = k >= 16 && k <= 21;              // inatm reads the cloud arrays only when icld >= 1
        if (!c.in[k] && !(cloud_arr && icld_eff == 0)) return fail(RRTMG_LW_HIP_EARG, "null input array (argument %d)", k);
    }
    for (int k = 0; k < 6; k++) if (!c.out[k]) return fail(RRTMG_LW_HIP_EARG, "null output array");
    if (Q.idrv == 1 && (!duflx_dt || !duflxc_dt)) return fail(RRTMG_LW_HIP_EARG, "idrv=1 needs duflx_dt and duflxc_dt");
    if (Q.ncol + ncol > 0x7fffffffLL) return fail(RRTMG_LW_HIP_EARG, "too many queued columns");
    Q.chunks.push_back(c);
    Q.ncol += ncol;
    return 0;
}

}   // extern "C"

// f(i) for i in [0, n) on up to 8 host threads (contiguous index ranges)
template <class F>
static void queue_parallel(size_t n, F f)
{
    // (on the persistent host threads: a pass of the combining entry packs sixteen chunks, and eight thread starts cost more than that)
    if (n < 4) { for (size_t i = 0; i < n; i++) f(i); return; }
    const size_t want = std::min<size_t>(n, 8);
    host_parallel(want << 20, [&](int t, int nt) { for (size_t i = n * (size_t)t / (size_t)nt; i < n * (size_t)(t + 1) / (size_t)nt; i++) f(i); });
}

// The chunks as ONE call: every array is [rows][columns][inner] - row r of a chunk goes to row r of the packed array at the chunk's column
// offset (pinned set Q.pinned) - then nomcica_host on the packed arrays, then the outputs back to every chunk.  Caller holds the entry lock.
// kind 0: rrtmg_lw (non-McICA); kind 1: the fused sub-column generator + McICA solver with the kissvec generator (a stream per column:
// the columns of a packed call draw what they draw on their own), permuteseed and - has_alpha - the generator's alpha array as input 23
static int mcica_subcol_host(int ncol, int nlay, int *icld, int idrv, int permuteseed, int *irng,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp, const double *reice,
    const double *reliq, const double *alpha, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc, double *duflx_dt, double *duflxc_dt);
static int solve_chunks(const std::vector<QueuedChunk> &chunks, long long N, int nlay, int icld_in, int idrv, int inflg, int iceflg, int liqflg,
                        int kind = 0, int permuteseed = 0, bool has_alpha = false)
{
    size_t in_inner[23], in_rows[23], out_rows[8];
    double *in_p[23], *out_p[8];
    queue_shapes(nlay, in_inner, in_rows, out_rows);
    size_t tot = 0;
    for (int k = 0; k < 23; k++) tot += in_inner[k] * in_rows[k] * (size_t)N;
    for (int k = 0; k < 8; k++) tot += out_rows[k] * (size_t)N;
    if (has_alpha) tot += (size_t)nlay * (size_t)N;
    if (Q.pinned_doubles < tot) {
        if (Q.pinned) { forget_pinned(Q.pinned); (void)hipHostFree(Q.pinned); Q.pinned = nullptr; Q.pinned_doubles = 0; }
        HIP_TRY(hipHostMalloc((void **)&Q.pinned, tot * sizeof(double), hipHostMallocDefault));
        Q.pinned_doubles = tot;
        g_pinned.emplace_back((const char *)Q.pinned, tot * sizeof(double));
    }
    double *p = Q.pinned;
    for (int k = 0; k < 23; k++) { in_p[k] = p; p += in_inner[k] * in_rows[k] * (size_t)N; }
    for (int k = 0; k < 8; k++) { out_p[k] = p; p += out_rows[k] * (size_t)N; }
    double *alpha_p = has_alpha ? p : nullptr;
    const bool cloud = !(icld_in == 0);
    std::vector<size_t> offs(chunks.size());
    { size_t off = 0; for (size_t i = 0; i < chunks.size(); i++) { offs[i] = off; off += (size_t)chunks[i].ncol; } }
    queue_parallel(chunks.size(), [&](size_t i) {          // (the copies are many short rows: host-bound, spread over a few threads)
        const QueuedChunk &c = chunks[i];
        for (int k = 0; k < 23; k++) {
            const bool cloud_arr = k >= 16 && k <= 21;
            if (cloud_arr && !cloud) continue;
            const size_t w = in_inner[k] * (size_t)c.ncol;
            for (size_t r = 0; r < in_rows[k]; r++)
                memcpy(in_p[k] + (r * (size_t)N + offs[i]) * in_inner[k], c.in[k] + r * w, w * sizeof(double));
        }
        if (has_alpha && cloud)
            for (size_t r = 0; r < (size_t)nlay; r++) memcpy(alpha_p + r * (size_t)N + offs[i], c.in[23] + r * (size_t)c.ncol, (size_t)c.ncol * sizeof(double));
    });
    int icld = icld_in, irng = 0;
    const int rc = kind == 1
        ? mcica_subcol_host((int)N, nlay, &icld, idrv, permuteseed, &irng, in_p[0], in_p[1], in_p[2], in_p[3], in_p[4], in_p[5], in_p[6], in_p[7],
                            in_p[8], in_p[9], in_p[10], in_p[11], in_p[12], in_p[13], in_p[14], in_p[15], inflg, iceflg, liqflg,
                            in_p[16], in_p[17], in_p[18], in_p[19], in_p[20], in_p[21], alpha_p, in_p[22],
                            out_p[0], out_p[1], out_p[2], out_p[3], out_p[4], out_p[5], idrv == 1 ? out_p[6] : nullptr, idrv == 1 ? out_p[7] : nullptr)
        : nomcica_host((int)N, nlay, &icld, idrv, in_p[0], in_p[1], in_p[2], in_p[3], in_p[4], in_p[5], in_p[6], in_p[7],
                                            in_p[8], in_p[9], in_p[10], in_p[11], in_p[12], in_p[13], in_p[14], in_p[15], inflg, iceflg, liqflg,
                                            cloud ? in_p[16] : nullptr, cloud ? in_p[17] : nullptr, cloud ? in_p[18] : nullptr, cloud ? in_p[19] : nullptr,
                                            cloud ? in_p[20] : nullptr, cloud ? in_p[21] : nullptr, in_p[22],
                                            out_p[0], out_p[1], out_p[2], out_p[3], out_p[4], out_p[5],
                                            idrv == 1 ? out_p[6] : nullptr, idrv == 1 ? out_p[7] : nullptr);
    if (rc == 0) {
        queue_parallel(chunks.size(), [&](size_t i) {
            const QueuedChunk &c = chunks[i];
            for (int k = 0; k < 8; k++) {
                if (!c.out[k] || (k >= 6 && idrv != 1)) continue;
                for (size_t r = 0; r < out_rows[k]; r++)
                    memcpy(c.out[k] + r * (size_t)c.ncol, out_p[k] + r * (size_t)N + offs[i], (size_t)c.ncol * sizeof(double));
            }
            if (c.icld) *c.icld = icld;
        });
    }
    return rc;
}

extern "C" int rrtmg_lw_hip_queue_flush(void)
{
    ENTRY_LOCK;             // held over pack, solve and scatter: another thread's queue_add / queue_begin / finalize must not touch the chunk list or the pinned set in between
    if (!Q.open) return fail(RRTMG_LW_HIP_EARG, "rrtmg_lw_hip_queue_begin has not been called");
    if (Q.ncol == 0) return 0;
    const int rc = solve_chunks(Q.chunks, Q.ncol, Q.nlay, Q.icld, Q.idrv, Q.inflg, Q.iceflg, Q.liqflg);
    Q.chunks.clear();
    Q.ncol = 0;
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// Concurrent callers (threading): a host model that calls rrtmg_lw per chunk of columns from several OpenMP threads.
// The reference is serial inside a call and hosts thread OVER calls (src/rrtmg_lw_rad.nomcica.f90:472); here a call costs ~0.9 ms
// whatever its size below a few thousand columns, so N threads taking turns at one lock run at one chunk per 0.9 ms.  Instead a caller
// that finds another call in flight leaves its request in a list and sleeps; whoever holds the turn solves everything that has gathered
// with the same (nlay, icld, idrv, cloud flags) as ONE packed call (solve_chunks, the queue's own code) and wakes the owners.  Results do
// not depend on the company a chunk had (tests/test_hip_parity.py::test_a_column_does_not_depend_on_its_neighbours).  No call-site change.
// ---------------------------------------------------------------------------------------------------
namespace {
struct CallReq {
    QueuedChunk c;
    int nlay, icld, idrv, inflg, iceflg, liqflg;
    int kind = 0, permuteseed = 0;             // kind 1: the fused generator + McICA entry (kissvec); see solve_chunks
    int rc = 0;
    bool done = false, lead = false;
    std::string err;                           // the text of rc != 0, handed to the owner's thread
    std::condition_variable cv;
};
std::mutex g_comb_mu;
std::condition_variable g_comb_arrive;      // a request has joined the list
std::vector<CallReq *> g_comb_pending;
bool g_comb_busy = false;
size_t g_comb_last = 1;                      // requests the previous pass served: so many callers are about to come back
std::atomic<long long> g_comb_calls{0}, g_comb_passes{0};
// calls of up to this many columns are candidates (larger ones fill the device on their own and go straight to the entry lock)
const int COMBINE_MAX = []() { const char *e = getenv("RRTMG_LW_COMBINE_MAX"); return e ? atoi(e) : 8192; }();
const int COMBINE_WAIT_US = []() { const char *e = getenv("RRTMG_LW_COMBINE_WAIT_US"); return e ? atoi(e) : 250; }();

// everything that has gathered, group by group (the first request's shape, then what is left, ...)
void comb_serve(std::vector<CallReq *> &batch)
{
    ENTRY_LOCK;
    std::vector<char> served(batch.size(), 0);
    for (size_t i = 0; i < batch.size(); i++) {
        if (served[i]) continue;
        CallReq &a = *batch[i];
        std::vector<size_t> grp;
        long long N = 0;
        for (size_t j = i; j < batch.size(); j++) {
            const CallReq &b = *batch[j];
            if (!served[j] && b.nlay == a.nlay && b.icld == a.icld && b.idrv == a.idrv && b.inflg == a.inflg && b.iceflg == a.iceflg && b.liqflg == a.liqflg &&
                b.kind == a.kind && b.permuteseed == a.permuteseed && (b.c.in[23] != nullptr) == (a.c.in[23] != nullptr) &&
                N + b.c.ncol <= 0x7fffffffLL) { grp.push_back(j); N += b.c.ncol; }
        }
        auto alone = [&](CallReq &r) {
            const QueuedChunk &c = r.c;
            int icld = r.icld, irng = 0;
            g_comb_passes++;
            if (r.kind == 1)
                r.rc = mcica_subcol_host(c.ncol, r.nlay, &icld, r.idrv, r.permuteseed, &irng, c.in[0], c.in[1], c.in[2], c.in[3], c.in[4], c.in[5], c.in[6], c.in[7],
                                         c.in[8], c.in[9], c.in[10], c.in[11], c.in[12], c.in[13], c.in[14], c.in[15], r.inflg, r.iceflg, r.liqflg, c.in[16], c.in[17],
                                         c.in[18], c.in[19], c.in[20], c.in[21], c.in[23], c.in[22], c.out[0], c.out[1], c.out[2], c.out[3], c.out[4], c.out[5], c.out[6], c.out[7]);
            else
            r.rc = nomcica_host(c.ncol, r.nlay, &icld, r.idrv, c.in[0], c.in[1], c.in[2], c.in[3], c.in[4], c.in[5], c.in[6], c.in[7], c.in[8], c.in[9], c.in[10],
                                c.in[11], c.in[12], c.in[13], c.in[14], c.in[15], r.inflg, r.iceflg, r.liqflg, c.in[16], c.in[17], c.in[18], c.in[19], c.in[20], c.in[21],
                                c.in[22], c.out[0], c.out[1], c.out[2], c.out[3], c.out[4], c.out[5], c.out[6], c.out[7]);
            if (c.icld) *c.icld = icld;
            if (r.rc != 0) r.err = G.err;
        };
        if (grp.size() == 1) alone(a);
        else {
            std::vector<QueuedChunk> chunks;
            for (size_t j : grp) chunks.push_back(batch[j]->c);
            g_comb_passes++;
            const int rc = solve_chunks(chunks, N, a.nlay, a.icld, a.idrv, a.inflg, a.iceflg, a.liqflg, a.kind, a.permuteseed, a.c.in[23] != nullptr);
            if (rc == 0) { for (size_t j : grp) batch[j]->rc = 0; }
            else { for (size_t j : grp) alone(*batch[j]); }      // an error (one caller's bad particle size ...) belongs to the call that caused it: each chunk again, on its own
        }
        for (size_t j : grp) served[j] = 1;
    }
}

int comb_call(CallReq &me)
{
    std::unique_lock<std::mutex> lk(g_comb_mu);
    g_comb_calls++;
    g_comb_pending.push_back(&me);
    g_comb_arrive.notify_one();
    if (g_comb_busy) {
        me.cv.wait(lk, [&] { return me.done || me.lead; });
        if (me.done) { if (me.rc != 0) tl_err = me.err; return me.rc; }
    }
    g_comb_busy = true;                      // this thread has the turn: its own request is among the pending ones
    while (!me.done) {
        // The callers the previous pass served come back within the time of a call's return trip: a pass that starts the moment the first of
        // them is here serves that one alone while the others queue up behind it (measured: passes of 1 and 15 chunks in turn).  Wait for as
        // many as there were, a fraction of a pass's own time at most; a single-threaded host (previous pass: one request) never waits.
        // The window adapts: a wait that ended with everybody here keeps (or restores) it, one that ran out halves it - two threads of
        // which one comes back only after ten milliseconds of its own work must not cost the other 250 us per call - and every 64th pass
        // tries a short window again.
        if (g_comb_last > 1) {
            static int window_us = COMBINE_WAIT_US;
            static unsigned npass = 0;
            int w = window_us;
            if (w < 20 && (++npass & 63u) == 0u) w = std::min(60, COMBINE_WAIT_US);
            if (w >= 20) {
                const bool all = g_comb_arrive.wait_for(lk, std::chrono::microseconds(w), [&] { return g_comb_pending.size() >= g_comb_last; });
                window_us = all ? std::min(COMBINE_WAIT_US, std::max(60, 2 * w)) : w / 2;
            }
        }
        std::vector<CallReq *> batch;
        batch.swap(g_comb_pending);
        g_comb_last = batch.size();
        lk.unlock();
        {   // (the chunks of other callers go through fail() on this thread: their texts are theirs - CallReq::err - not this thread's)
            const std::string mine = tl_err;
            comb_serve(batch);
            tl_err = mine;
        }
        lk.lock();
        for (CallReq *r : batch) { r->done = true; if (r != &me) r->cv.notify_one(); }
    }
    if (!g_comb_pending.empty()) { g_comb_pending.front()->lead = true; g_comb_pending.front()->cv.notify_one(); }     // the turn goes to the first of those who came meanwhile
    else g_comb_busy = false;
    if (me.rc != 0) tl_err = me.err;
    return me.rc;
}

int comb_max() { return COMBINE_MAX; }
bool comb_enabled() { static const bool on = []() { const char *e = getenv("RRTMG_LW_COMBINE"); return !e || atoi(e) != 0; }(); return on; }
long long comb_calls_total() { return g_comb_calls.load(); }
long long comb_passes_total() { return g_comb_passes.load(); }

int nomcica_combined(int ncol, int nlay, int *icld, int idrv,
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr,
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,
    const double *ccl4vmr, const double *emis, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
    const double *reice, const double *reliq, const double *tauaer,
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
    double *duflx_dt, double *duflxc_dt)
{
    CallReq me;
    me.c = QueuedChunk{ncol, icld,
                       {play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis,
                        cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer},
                       {uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt}};
    me.nlay = nlay; me.icld = *icld; me.idrv = idrv; me.inflg = inflglw; me.iceflg = iceflglw; me.liqflg = liqflglw;
    // what a packed call could not report per chunk is checked here, per call: null arrays (the packing would read them)
    const int icld_eff = (me.icld < 0 || me.icld > 3) ? 2 : me.icld;
    bool bad = !uflx || !dflx || !hr || !uflxc || !dflxc || !hrc || (idrv == 1 && (!duflx_dt || !duflxc_dt)) || nlay < 1 || nlay > 603;
    for (int k = 0; k < 23; k++) {
        const bool cloud_arr = k >= 16 && k <= 21;
        if (!me.c.in[k] && !(cloud_arr && icld_eff == 0)) bad = true;
    }
    if (bad) {          // let the plain entry say what is wrong
        ENTRY_LOCK;
        return nomcica_host(ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis,
                            inflglw, iceflglw, liqflglw, cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt);
    }
    return comb_call(me);
}
}   // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------------
// McICA flavour
// ---------------------------------------------------------------------------------------------------
#define GCM_PARAMS                                                                                              \
    const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,         \
    const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr, const double *n2ovmr, \
    const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr, const double *cfc22vmr,               \
    const double *ccl4vmr, const double *emis
#define OUT_PARAMS                                                                                              \
    double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc, double *duflx_dt, double *duflxc_dt

int rrtmg_lw_hip_run_mcica_device(
    int ncol, int nlay, int *icld, int idrv, GCM_PARAMS, int inflglw, int iceflglw, int liqflglw,
    const double *cldfmcl, const double *taucmcl, const double *ciwpmcl, const double *clwpmcl,
    const double *reicmcl, const double *relqmcl, const double *tauaer, OUT_PARAMS, void *stream)
{
    ENTRY_LOCK_FOR(play);
    if (int rc = check_mcica_build()) return rc;
    if (int rc = check_common(ncol, nlay)) return rc;
    if (!icld) return fail(RRTMG_LW_HIP_EARG, "icld is null");
    if (*icld < 0 || *icld > 3) *icld = 2;                       // src/rrtmg_lw_rad.f90:469
    if (idrv == 1 && (!duflx_dt || !duflxc_dt)) return fail(RRTMG_LW_HIP_EARG, "idrv=1 needs duflx_dt and duflxc_dt");
    const int mode = *icld == 0 ? 0 : 3;                         // inatm leaves the cloud arrays zero when icld = 0 (:899-911)
    const int nbmax = balanced_batch(ncol, eff_batch(nlay));
    if (int rc = ensure_workspace(nlay, nbmax, mode != 0, mode == 3, idrv, mode)) return rc;
    GcmIn g{play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr,
            ccl4vmr, emis, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, tauaer};
    McIn m{cldfmcl, taucmcl, ciwpmcl, clwpmcl, reicmcl, relqmcl};
    FluxOut out{uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt, nullptr, nullptr};
    return run_pipelined((hipStream_t)stream, ncol, nlay, mode, idrv, g, inflglw, iceflglw, liqflglw, out, &m);
}

int rrtmg_lw_hip_run_mcica(
    int ncol, int nlay, int *icld, int idrv, GCM_PARAMS, int inflglw, int iceflglw, int liqflglw,
    const double *cldfmcl, const double *taucmcl, const double *ciwpmcl, const double *clwpmcl,
    const double *reicmcl, const double *relqmcl, const double *tauaer, OUT_PARAMS)
{
    ENTRY_LOCK;
    if (int rc = check_mcica_build()) return rc;
    if (int rc = check_common(ncol, nlay)) return rc;
    if (!icld) return fail(RRTMG_LW_HIP_EARG, "icld is null");
    if (*icld < 0 || *icld > 3) *icld = 2;
    if (idrv == 1 && (!duflx_dt || !duflxc_dt)) return fail(RRTMG_LW_HIP_EARG, "idrv=1 needs duflx_dt and duflxc_dt");
    const int mode = *icld == 0 ? 0 : 3;
    const bool cloud = mode == 3;
    return fan_out(ncol, [&](int c0, int c1) -> int {     // (one block of columns per device of rrtmg_lw_hip_init_devices)
    if (int rc = check_common(ncol, nlay)) return rc;
    HIP_TRY(hipDeviceSynchronize());        // asynchronous device-entry work of earlier calls shares the workspace
    // the sub-column arrays are 4 x 140 x nlay doubles per column: bound the batch so that staging stays below ~4 GB
    const int mcmax = (int)std::max<size_t>(64, ((size_t)1 << 30) / ((size_t)4 * NGPT * nlay * 8));        // (HOST_SETS staging sets of 1 GiB)
    const int nbmax = balanced_batch(c1 - c0, std::min(std::min(eff_batch(nlay), HOST_BATCH), cloud ? mcmax : HOST_BATCH));
    if (int rc = ensure_workspace(nlay, nbmax, cloud, cloud, idrv, mode)) return rc;
    const size_t L = (size_t)nlay;
    std::vector<HostIn> ins = {
        {play, 1, L, 0}, {plev, 1, L + 1, 0}, {tlay, 1, L, 0}, {tlev, 1, L + 1, 0}, {tsfc, 1, 1, 0},
        {h2ovmr, 1, L, 0}, {o3vmr, 1, L, 0}, {co2vmr, 1, L, 0}, {ch4vmr, 1, L, 0}, {n2ovmr, 1, L, 0}, {o2vmr, 1, L, 0},
        {cfc11vmr, 1, L, 0}, {cfc12vmr, 1, L, 0}, {cfc22vmr, 1, L, 0}, {ccl4vmr, 1, L, 0}, {emis, 1, 16, 0}, {tauaer, 1, 16 * L, 0},
        {cloud ? cldfmcl : nullptr, NGPT, L, 0}, {cloud ? taucmcl : nullptr, NGPT, L, 0}, {cloud ? ciwpmcl : nullptr, NGPT, L, 0},
        {cloud ? clwpmcl : nullptr, NGPT, L, 0}, {cloud ? reicmcl : nullptr, 1, L, 0}, {cloud ? relqmcl : nullptr, 1, L, 0}};
    for (size_t k = 0; k < 17; k++) if (!ins[k].h) return fail(RRTMG_LW_HIP_EARG, "null input array (argument %d)", (int)k);
    if (cloud) for (size_t k = 17; k < ins.size(); k++) if (!ins[k].h) return fail(RRTMG_LW_HIP_EARG, "null McICA cloud array");
    std::vector<HostOut> outs = {{uflx, L + 1, 0, true}, {dflx, L + 1, 0, true}, {hr, L, 0, true}, {uflxc, L + 1, 0, true},
                                 {dflxc, L + 1, 0, true}, {hrc, L, 0, true}, {duflx_dt, L + 1, 0, idrv == 1}, {duflxc_dt, L + 1, 0, idrv == 1}};
    auto body = [&](hipStream_t s, int nb, int, std::vector<HostIn> &in, std::vector<HostOut> &out_) -> int {
        GcmIn g{in[0].d, in[1].d, in[2].d, in[3].d, in[4].d, in[5].d, in[6].d, in[7].d, in[8].d, in[9].d, in[10].d,
                in[11].d, in[12].d, in[13].d, in[14].d, in[15].d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, in[16].d};
        McIn m{in[17].d, in[18].d, in[19].d, in[20].d, in[21].d, in[22].d};
        ColIn c{};
        FluxOut out{out_[0].d, out_[1].d, out_[2].d, out_[3].d, out_[4].d, out_[5].d, out_[6].d, out_[7].d, nullptr, nullptr};
        return run_batch<true>(s, nb, 0, nb, nlay, mode, idrv, 1, 16, g, c, inflglw, iceflglw, liqflglw, out, &m);
    };
    // layers whose sub-column cloud fractions are all below cldmin for the batch: cldprmc reads nothing else of them (src/rrtmg_lw_cldprmc.f90:182-183)
    std::vector<unsigned char> cloudfree(L, 0), cf_uni(L, 0);
    std::vector<uint64_t> cf_bits(L, 0);
    auto prep = [&](int, int col0, int nb, hipStream_t) -> int {
        if (!cloud) return 0;
        rows_below(cldfmcl, NGPT, L, (size_t)ncol, (size_t)col0, (size_t)nb, 1.e-20, cloudfree.data(), cf_uni.data(), cf_bits.data());
        ins[17].known = cf_uni.data(); ins[17].known_bits = cf_bits.data();        // (the cloud-free layers of cldfmcl have just been read to their end)
        for (size_t a = 18; a < ins.size(); a++) ins[a].skip = cloudfree.data();
        return 0;
    };
    if (int rc = host_pipeline(ncol, c0, c1, nbmax, ins, outs, body, prep)) return rc;
    hipStream_t s = G.stream;
    return read_physics_error(s);
    });
}

int rrtmg_lw_hip_get_alpha(int ncol, int nlay, int icld, int idcor, double decorr_con, const double *dz, const double *lat,
                           int juldat, const double *cldfrac, double *alpha)
{
    ENTRY_LOCK;
    if (int rc = check_common(ncol, nlay)) return rc;
    if (G.init) HIP_TRY(hipDeviceSynchronize());      // asynchronous device-entry work of earlier calls shares the workspace
    if (!(icld == 4 || icld == 5)) return 0;                      // alpha is only defined for the exponential overlaps
    const size_t n = (size_t)ncol, L = (size_t)nlay;
    std::vector<HostIn> ins = {{dz, 1, L, 0}, {lat, 1, 1, 0}, {cldfrac, 1, L, 0}};
    for (auto &a : ins) if (!a.h) return fail(RRTMG_LW_HIP_EARG, "null input array");
    std::vector<HostOut> outs = {{alpha, L, 0, true}};
    if (int rc = stage_alloc(ins, outs, n)) return rc;
    hipStream_t s = G.stream;
    if (int rc = stage_in(ins, n, 0, n, s)) return rc;
    const dim3 grid((ncol + BLOCK - 1) / BLOCK, nlay), block(BLOCK);
    hipLaunchKernelGGL(k_alpha, grid, block, 0, s, ncol, nlay, icld, idcor, decorr_con, (const double *)ins[0].d, (const double *)ins[1].d,
                       juldat, (const double *)ins[2].d, outs[0].d);
    return stage_out(outs, n, 0, n, s);
}

int rrtmg_lw_hip_mcica_subcol_device(
    int ncol, int nlay, int icld, int permuteseed, int *irng, const double *play, const double *cldfrac, const double *ciwp,
    const double *clwp, const double *rei, const double *rel, const double *tauc, const double *alpha,
    double *cldfmcl, double *ciwpmcl, double *clwpmcl, double *reicmcl, double *relqmcl, double *taucmcl, void *stream)
{
    ENTRY_LOCK_FOR(play, irng && *irng != 0);
    if (int rc = check_mcica_build()) return rc;
    if (int rc = check_subcol_args(ncol, nlay, icld, irng)) return rc;
    if (icld == 0) return 0;                                      // src/mcica_subcol_gen_lw.f90:265
    hipStream_t s = (hipStream_t)stream;
    if (int rc = generate_mask(s, ncol, nlay, icld, permuteseed, *irng, play, cldfrac, alpha)) return rc;
    const size_t tot = (size_t)NGPT * ncol;
    const dim3 grid((unsigned)((tot + BLOCK - 1) / BLOCK), nlay), block(BLOCK);
    LAUNCH("k_subcol_expand", k_subcol_expand, grid, block, s, G.W, ciwp, clwp, tauc, ncol, nlay, 0, cldfmcl, ciwpmcl, clwpmcl, taucmcl);
    HIP_TRY(hipMemcpyAsync(reicmcl, rei, (size_t)ncol * nlay * 8, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(relqmcl, rel, (size_t)ncol * nlay * 8, hipMemcpyDeviceToDevice, s));
    return 0;
}

// columns [c0, c1) of the stand-alone generator with host arrays (ncol columns wide) on the calling thread's current device state
static int subcol_host_range(
    int ncol, int c0, int c1, int nlay, int icld, int permuteseed, int irng, const double *play, const double *cldfrac, const double *ciwp,
    const double *clwp, const double *rei, const double *rel, const double *tauc, const double *alpha,
    double *cldfmcl, double *ciwpmcl, double *clwpmcl, double *reicmcl, double *relqmcl, double *taucmcl)
{
    HIP_TRY(hipDeviceSynchronize());      // asynchronous device-entry work of earlier calls shares the workspace
    const size_t n = (size_t)ncol, L = (size_t)nlay, nl = (size_t)(c1 - c0);
    const bool two = icld == 4 || icld == 5;
    // (only what the generator reads travels: pressures, cloud fractions, alpha; water paths and optical depths stay with the host threads below)
    std::vector<HostIn> ins = {{play, 1, L, 0}, {cldfrac, 1, L, 0}, {two ? alpha : nullptr, 1, L, 0}};
    std::vector<HostOut> outs;                                   // outputs are written by the host threads below
    if (int rc = stage_alloc(ins, outs, nl)) return rc;
    hipStream_t s = G.stream;
    if (int rc = stage_in(ins, n, (size_t)c0, nl, s)) return rc;
    if (int rc = generate_mask(s, (int)nl, nlay, icld, permuteseed, irng, ins[0].d, ins[1].d, ins[2].d)) return rc;
    // The sub-column arrays are implied by the mask (MASK_WORDS x 4 B per (column, layer)) and the grid-mean inputs, which the host holds:
    // the mask comes back over PCIe and the host threads write the four (ngpt, ncol, nlay) arrays - 4 x ngpt x 8 B per (column, layer), 224
    // times the mask - at memory speed (src/mcica_subcol_gen_lw.f90:664-680: where cloudy the layer's water paths and the band's optical
    // depth, elsewhere zero).  (Expanding on the device and copying 322 KB per 72-layer column back ran at 0.03 M columns/s.)
    std::vector<unsigned> hmask((size_t)MASK_WORDS * L * nl);
    HIP_TRY(hipStreamSynchronize(s));
    if (int rc = bounce_d2h(hmask.data(), G.mask, hmask.size() * sizeof(unsigned))) return rc;
    int gband[NGPT];
    for (int ig = 0; ig < NGPT; ig++) { int b = 1; for (int B = 2; B <= NBND; B++) b += (ig >= band_g0(B)) ? 1 : 0; gband[ig] = b - 1; }
    host_parallel((size_t)NGPT * nl * L * 8 * 4, [&](int t, int nt) {
        const size_t cells = nl * L;
        for (size_t ci = cells * (size_t)t / (size_t)nt; ci < cells * (size_t)(t + 1) / (size_t)nt; ci++) {
            const size_t l = ci / nl, lc = ci % nl;                    // the block's (column, layer) order
            const size_t cl = (size_t)c0 + lc + n * l;                 // ... and the cell's place in the caller's arrays
            unsigned w[MASK_WORDS];
            for (int k = 0; k < MASK_WORDS; k++) w[k] = hmask[((size_t)k * L + l) * nl + lc];
            const double iw = ciwp[cl], lw = clwp[cl];
            const double *tc = tauc + (size_t)NBND * cl;
            double *o0 = cldfmcl + (size_t)NGPT * cl, *o1 = ciwpmcl + (size_t)NGPT * cl, *o2 = clwpmcl + (size_t)NGPT * cl, *o3 = taucmcl + (size_t)NGPT * cl;
            for (int ig = 0; ig < NGPT; ig++) {
                const bool on = (w[ig >> 5] >> (ig & 31)) & 1u;
                o0[ig] = on ? 1.0 : 0.0;
                o1[ig] = on ? iw : 0.0;
                o2[ig] = on ? lw : 0.0;
                o3[ig] = on ? tc[gband[ig]] : 0.0;
            }
        }
    });
    for (size_t l = 0; l < L; l++) {                            // :283-284
        memcpy(reicmcl + c0 + n * l, rei + c0 + n * l, nl * 8);
        memcpy(relqmcl + c0 + n * l, rel + c0 + n * l, nl * 8);
    }
    return read_physics_error(s);
}

int rrtmg_lw_hip_mcica_subcol(
    int ncol, int nlay, int icld, int permuteseed, int *irng, const double *play, const double *cldfrac, const double *ciwp,
    const double *clwp, const double *rei, const double *rel, const double *tauc, const double *alpha,
    double *cldfmcl, double *ciwpmcl, double *clwpmcl, double *reicmcl, double *relqmcl, double *taucmcl)
{
    ENTRY_LOCK;
    if (int rc = check_mcica_build()) return rc;
    if (int rc = check_subcol_args(ncol, nlay, icld, irng)) return rc;
    if (icld == 0) return 0;
    if (!play || !cldfrac || !ciwp || !clwp || !rei || !rel || !tauc) return fail(RRTMG_LW_HIP_EARG, "null input array");
    if ((icld == 4 || icld == 5) && !alpha) return fail(RRTMG_LW_HIP_EARG, "icld = 4/5 needs alpha");
    if (!cldfmcl || !ciwpmcl || !clwpmcl || !reicmcl || !relqmcl || !taucmcl) return fail(RRTMG_LW_HIP_EARG, "null output array");
    const int rng = *irng;
    auto range = [&](int c0, int c1) -> int {
        return subcol_host_range(ncol, c0, c1, nlay, icld, permuteseed, rng, play, cldfrac, ciwp, clwp, rei, rel, tauc, alpha,
                                 cldfmcl, ciwpmcl, clwpmcl, reicmcl, relqmcl, taucmcl);
    };
    if (rng != 0) return range(0, ncol);          // one Mersenne-Twister stream over all columns (:497-503): the first device draws it
    return fan_out(ncol, range);                  // kissvec: a stream per column (:463-474)
}

// Fused generator + solver: mcica_subcol_lw followed by the McICA rrtmg_lw without materialising the (140,ncol,nlay)
// sub-column arrays (they are implied by the mask and the grid-mean cloud properties).  DEVICE pointers.
int rrtmg_lw_hip_run_mcica_subcol_device(
    int ncol, int nlay, int *icld, int idrv, int permuteseed, int *irng, GCM_PARAMS, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp, const double *reice,
    const double *reliq, const double *alpha, const double *tauaer, OUT_PARAMS, void *stream)
{
    ENTRY_LOCK_FOR(play, irng && *irng != 0);
    if (int rc = check_mcica_build()) return rc;
    if (!icld) return fail(RRTMG_LW_HIP_EARG, "icld is null");
    if (int rc = check_subcol_args(ncol, nlay, *icld, irng)) return rc;
    if (idrv == 1 && (!duflx_dt || !duflxc_dt)) return fail(RRTMG_LW_HIP_EARG, "idrv=1 needs duflx_dt and duflxc_dt");
    const int icld_gen = *icld;
    if (*icld > 3) *icld = 2;                                     // what rrtmg_lw does to the generator's icld (src/rrtmg_lw_rad.f90:469)
    const int mode = icld_gen == 0 ? 0 : 3;
    hipStream_t s = (hipStream_t)stream;
    const int nbmax = balanced_batch(ncol, eff_batch(nlay));
    if (int rc = ensure_workspace(nlay, nbmax, mode != 0, false, idrv, mode)) return rc;      // mask path: no per-g-point cloud arrays
    KissGen gen{false, icld_gen, permuteseed, alpha};
    if (mode == 3) {
        if (*irng == 0) {          // kissvec: every column owns its stream -> generated batch by batch on the auxiliary stream
            if (int rc = prepare_mask(ncol, nlay, icld_gen, 0, alpha)) return rc;
            gen.on = true;
        } else if (int rc = generate_mask(s, ncol, nlay, icld_gen, permuteseed, *irng, play, cldfr, alpha)) return rc;
    }
    GcmIn g{play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr,
            ccl4vmr, emis, cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer};
    FluxOut out{uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt, nullptr, nullptr};
    return run_pipelined(s, ncol, nlay, mode, idrv, g, inflglw, iceflglw, liqflglw, out, nullptr, gen);
}

// columns [c0, c1) of the fused generator + solver call on the calling thread's current device state (kissvec seeds its stream per
// column, src/mcica_subcol_gen_lw.f90:463-474: a block of columns is generated and solved on its own; the Mersenne-Twister stream is one
// sequence over all columns of a call, :497-503, and is drawn with c0 = 0, c1 = ncol only)
static int mcica_subcol_host_range(
    int ncol, int c0, int c1, int nlay, int icld_gen, int idrv, int permuteseed, int irng, GCM_PARAMS, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp, const double *reice,
    const double *reliq, const double *alpha, const double *tauaer, OUT_PARAMS)
{
    if (int rc = check_common(ncol, nlay)) return rc;
    HIP_TRY(hipDeviceSynchronize());      // asynchronous device-entry work of earlier calls shares the workspace
    const int mode = icld_gen == 0 ? 0 : 3;
    const bool cloud = mode == 3, two = icld_gen == 4 || icld_gen == 5;
    const int nloc = c1 - c0;
    const int nbmax = balanced_batch(nloc, std::min(eff_batch(nlay), HOST_BATCH));
    if (int rc = ensure_workspace(nlay, nbmax, cloud, false, idrv, mode)) return rc;
    const size_t L = (size_t)nlay, n = (size_t)ncol, nl = (size_t)nloc;
    hipStream_t s = G.stream;
    // 1. masks of the block's columns: needs play, cldfr, alpha of every one of them
    double *d_gen = nullptr;
    if (cloud) {
        if (!play || !cldfr || (two && !alpha)) return fail(RRTMG_LW_HIP_EARG, "null generator input");
        HIP_TRY(hipMalloc((void **)&d_gen, nl * L * 8 * 3));
        // (through the library's own pinned buffer: see bounce_h2d)
        int rc = bounce_h2d_rows(d_gen, play + c0, nl * 8, n * 8, L);
        if (rc == 0) rc = bounce_h2d_rows(d_gen + nl * L, cldfr + c0, nl * 8, n * 8, L);
        if (rc == 0 && two) rc = bounce_h2d_rows(d_gen + 2 * nl * L, alpha + c0, nl * 8, n * 8, L);
        if (rc == 0) rc = generate_mask(s, nloc, nlay, icld_gen, permuteseed, irng, d_gen, d_gen + nl * L, two ? d_gen + 2 * nl * L : nullptr);
        if (rc == 0 && hipStreamSynchronize(s) != hipSuccess) rc = fail(RRTMG_LW_HIP_EHIP, "generator failed");
        (void)hipFree(d_gen);
        if (rc) return rc;
    }
    // 2. column batches
    std::vector<HostIn> ins = {
        {play, 1, L, 0}, {plev, 1, L + 1, 0}, {tlay, 1, L, 0}, {tlev, 1, L + 1, 0}, {tsfc, 1, 1, 0},
        {h2ovmr, 1, L, 0}, {o3vmr, 1, L, 0}, {co2vmr, 1, L, 0}, {ch4vmr, 1, L, 0}, {n2ovmr, 1, L, 0}, {o2vmr, 1, L, 0},
        {cfc11vmr, 1, L, 0}, {cfc12vmr, 1, L, 0}, {cfc22vmr, 1, L, 0}, {ccl4vmr, 1, L, 0}, {emis, 1, 16, 0}, {tauaer, 1, 16 * L, 0},
        {cloud ? cldfr : nullptr, 1, L, 0}, {cloud ? taucld : nullptr, NBND, L, 0}, {cloud ? cicewp : nullptr, 1, L, 0},
        {cloud ? cliqwp : nullptr, 1, L, 0}, {cloud ? reice : nullptr, 1, L, 0}, {cloud ? reliq : nullptr, 1, L, 0}};
    for (size_t k = 0; k < 17; k++) if (!ins[k].h) return fail(RRTMG_LW_HIP_EARG, "null input array (argument %d)", (int)k);
    if (cloud) for (size_t k = 17; k < ins.size(); k++) if (!ins[k].h) return fail(RRTMG_LW_HIP_EARG, "null cloud array");
    std::vector<HostOut> outs = {{uflx, L + 1, 0, true}, {dflx, L + 1, 0, true}, {hr, L, 0, true}, {uflxc, L + 1, 0, true},
                                 {dflxc, L + 1, 0, true}, {hrc, L, 0, true}, {duflx_dt, L + 1, 0, idrv == 1}, {duflxc_dt, L + 1, 0, idrv == 1}};
    for (size_t k = 0; k < 6; k++) if (!outs[k].h) return fail(RRTMG_LW_HIP_EARG, "null output array");
    auto body = [&](hipStream_t bs, int nb, int col0, std::vector<HostIn> &in, std::vector<HostOut> &out_) -> int {
        GcmIn g{in[0].d, in[1].d, in[2].d, in[3].d, in[4].d, in[5].d, in[6].d, in[7].d, in[8].d, in[9].d, in[10].d,
                in[11].d, in[12].d, in[13].d, in[14].d, in[15].d, in[17].d, in[18].d, in[19].d, in[20].d, in[21].d, in[22].d, in[16].d};
        ColIn c{};
        FluxOut out{out_[0].d, out_[1].d, out_[2].d, out_[3].d, out_[4].d, out_[5].d, out_[6].d, out_[7].d, nullptr, nullptr};
        G.W.mask_col0 = (size_t)(col0 - c0);                      // staged arrays start at column 0, the mask holds the block's columns
        return run_batch<true>(bs, nb, 0, nb, nlay, mode, idrv, 1, 16, g, c, inflglw, iceflglw, liqflglw, out, nullptr);
    };
    {
        const int rc = host_pipeline(ncol, c0, c1, nbmax, ins, outs, body);
        if (rc) { G.W.mask_col0 = 0; return rc; }
    }
    G.W.mask_col0 = 0;
    return read_physics_error(s);
}

}   // extern "C"
// the fused entry behind the lock (the caller holds it): argument checks, the columns over the devices
static int mcica_subcol_host(int ncol, int nlay, int *icld, int idrv, int permuteseed, int *irng, GCM_PARAMS, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp, const double *reice,
    const double *reliq, const double *alpha, const double *tauaer, OUT_PARAMS)
{
    if (int rc = check_mcica_build()) return rc;
    if (!icld) return fail(RRTMG_LW_HIP_EARG, "icld is null");
    if (int rc = check_subcol_args(ncol, nlay, *icld, irng)) return rc;
    if (idrv == 1 && (!duflx_dt || !duflxc_dt)) return fail(RRTMG_LW_HIP_EARG, "idrv=1 needs duflx_dt and duflxc_dt");
    const int icld_gen = *icld, rng = *irng;
    if (*icld > 3) *icld = 2;
    auto range = [&](int c0, int c1) -> int {
        return mcica_subcol_host_range(ncol, c0, c1, nlay, icld_gen, idrv, permuteseed, rng, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr,
                                       n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw, cldfr, taucld, cicewp, cliqwp,
                                       reice, reliq, alpha, tauaer, uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt);
    };
    // the Mersenne-Twister stream (irng = 1) is ONE sequence over the (sub-column, column, layer) draws of the call
    // (src/mcica_subcol_gen_lw.f90:497-503): a column's deviates depend on every column before it, so the call stays on the first device
    if (rng != 0 && icld_gen != 0) return range(0, ncol);
    return fan_out(ncol, range);
}
extern "C" {
int rrtmg_lw_hip_run_mcica_subcol(
    int ncol, int nlay, int *icld, int idrv, int permuteseed, int *irng, GCM_PARAMS, int inflglw, int iceflglw, int liqflglw,
    const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp, const double *reice,
    const double *reliq, const double *alpha, const double *tauaer, OUT_PARAMS)
{
    // a small call with the kissvec generator (every column draws from its own stream: src/mcica_subcol_gen_lw.f90:463-474) that finds
    // another in flight is solved together with it in one pass, like the non-McICA entry (comb_call); the Mersenne Twister - one stream
    // over the columns of a call - and calls that are wrong on their face take the lock and say so themselves
    const bool plain = ncol >= 1 && ncol <= comb_max() && comb_enabled() && icld && irng && *irng == 0 && *icld >= 0 && *icld <= 5 && nlay >= 1 && nlay <= 603 &&
                       uflx && dflx && hr && uflxc && dflxc && hrc && (idrv != 1 || (duflx_dt && duflxc_dt)) &&
                       play && plev && tlay && tlev && tsfc && h2ovmr && o3vmr && co2vmr && ch4vmr && n2ovmr && o2vmr && cfc11vmr && cfc12vmr && cfc22vmr && ccl4vmr &&
                       emis && cldfr && taucld && cicewp && cliqwp && reice && reliq && tauaer;
    if (plain) {
        CallReq me;
        me.c = QueuedChunk{ncol, icld,
                           {play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis,
                            cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, alpha},
                           {uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt}};
        me.nlay = nlay; me.icld = *icld; me.idrv = idrv; me.inflg = inflglw; me.iceflg = iceflglw; me.liqflg = liqflglw;
        me.kind = 1; me.permuteseed = permuteseed;
        return comb_call(me);
    }
    ENTRY_LOCK;
    return mcica_subcol_host(ncol, nlay, icld, idrv, permuteseed, irng, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr,
                             cfc12vmr, cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw, cldfr, taucld, cicewp, cliqwp, reice, reliq, alpha, tauaer,
                             uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt);
}

}  // extern "C"
