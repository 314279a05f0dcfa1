// ekf_api.hip -- host side of libekfslam_hip.so: the C ABI of include/ekfslam_c.h.
//
// Owns the HBM layout (ekf_device.h), one HIP stream per handle, the immediate-mode input ring
// (host-mapped pinned memory, so an API call is kernel launches only), device-resident step
// scripts and their HIP-graph replay.  There is no CPU fallback: without a gfx950 device
// ekf_create fails with EKF_ERR_NO_DEVICE.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <string>
#include <atomic>
#include <mutex>
#include <functional>
#include <vector>

#include "ekf_device.h"

// single translation unit: the kernels are compiled together with their launch sites
#include "ekf_kernels.hip"
#include "ekf_solo.hip"

static thread_local std::string g_last_error;

static int set_error(int code, const char *what) {
    g_last_error = what ? what : "";
    return code;
}

int ekf_set_last_error(int code, const char *what) { return set_error(code, what); }  // for the library's other translation units (feat_api.hip)

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            char buf_[512];                                                                  \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return set_error(EKF_ERR_HIP, buf_);                                             \
        }                                                                                    \
    } while (0)

struct GraphEntry {
    int steps, M, has_truth;
    hipGraphExec_t exec;
};

struct ekf_batch {
    EkfDev dv;
    ekf_params params;
    int device;
    hipStream_t s_chain;  // chain kernels (and, without overlap, the dense passes too)
    hipStream_t s_flush;  // overlap: the dense passes; == s_chain otherwise
    bool overlap;         // a window's dense pass runs beside the next window's chain kernels (two slot sets, two Bm buffers)
    int prev_pending;     // overlap: slots of the other set whose dense pass has been launched but is not in Bm[buf_in]
    hipEvent_t ev_chain, ev_flush[2];
    hipEvent_t pass_done[2] = {nullptr, nullptr};  // completion of the pass that ev_flush[i] stands for: ev_flush[i] itself, or, while passes are profiled, that pass's stop event
    int ev_idx;           // ev_flush[ev_idx] belongs to the dense pass launched last
    bool chain_signalled; // the last chain launch carried ev_chain as its stop event
    int pass_seq;         // dense passes launched so far (overlap mode); k_mark stores it into dv.pass_flag behind each pass
    int need_pass;        // the pass the next chain launches have to wait for in-kernel, 0 = none
    bool inkernel_wait;   // chain kernels wait for their pass in-kernel (kernels of two streams run side by side), else by event
    long long chain_seq;  // chain launches so far; the kernel stores it into the host mirror when it is done
    int ncu = 0;                            // CUs of the device
    bool persist = true;                    // scripted runs in overlap mode: one chain launch for several windows (EKF_PERSIST=0 switches it off)
    unsigned long long seg_count_base[EKF_PLAN_MAX] = {};  // value dv.seg_count[i] reaches when every multi-segment launch enqueued so far has finished
    unsigned long long open_set_gate = 0;   // != 0: the open set was filled by segment open_gate_idx of a multi-segment launch: dv.seg_count[open_gate_idx] >= this opens its pass
    int open_gate_idx = 0;
    bool stats_in_mirror = false;  // mirror.stats is current (a chain launch ran since the last ekf_reset_stats)
    bool mirror_by_chain; // the newest writer of the host mirror is chain launch number chain_seq (else: some other kernel, synchronise)
    int flush_keep;       // pool key of s_flush: CUs kept free for the chain, -1 = unmasked
    bool solo = false;    // one workgroup per filter and one slot set (phase groups are possible)
    bool solo_fuse = false;    // ... and k_solo folds the windows it fills itself (ChainSeg::self_pass; EKF_SOLO_FUSE=0: k_flush_rb launches between the windows)
    long long prof_fused_passes = 0;  // dense passes executed inside profiled k_solo launches since the last ekf_flush_profile_read
    long long prof_solo_pairs = 0;    // ... and the number of those launches (event pairs that bracket a k_solo launch, not a k_flush_rb)
    bool solo_long = false;    // ... with a window longer than its own-row cache (k_solo<true>: the first half in accumulation registers)
    bool solo_kernel = false;  // ... run by k_solo (ekf_solo.hip: maps of up to 256 landmarks, one landmark per thread, one barrier per measurement)
    // Phase groups (solo batches): the filters are cut into ngroups ranges, each with a stream of its own on which its chain
    // launches and its dense passes alternate; the groups run out of phase, so that at any moment some groups are in their
    // (latency-bound) chain kernels while another streams its pass through HBM.  s_grp[0] == s_chain.
    int ngroups = 1;
    std::vector<hipStream_t> s_grp;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    int stagger_ticks = 0; // phase shift between consecutive groups at the start of a grouped run, in ticks of the 100 MHz clock
    int chain_wgs;        // k_chain workgroups per filter
    int chain_filters;    // filters per k_chain launch (all of the batch when its workgroups are resident together)
    int claimed_cus;      // CUs this handle's chain workgroups occupy when they run (residency registry, below)
    int solo_cus = 0;     // one-workgroup handles: CUs their launches occupy while they run (they claim none: g_cus_solo)
    bool flush_masked;    // s_flush is a dedicated CU-masked queue
    size_t chain_lds;     // dynamic LDS of a k_chain launch: the own-row cache
    bool chain_one = false;  // k_chain<true>: several workgroups per filter, at most one landmark per worker thread (EKF_CHAIN_ONE=0: the general kernel)
    double *bm1_base;     // allocation behind dv.Bm[1] (overlap mode)
    std::vector<int *> tile_maps;  // [nT]: XCD-aware wave -> tile tables of the row-block dense pass, built on demand
    bool xcd_map;         // EKF_XCD_MAP (default on)
    bool batch_interleave; // EKF_BATCH_INTERLEAVE (default on): batches run a filter's dense-pass workgroups on one XCD
    size_t device_bytes;
    int chain_threads;
    // host-side tracking
    int n_lm_hi;    // upper bound on max_b n_lm[b]
    int cur_set;    // slot set being filled
    int pending;    // slots used in cur_set
    int buf_in;     // Bm buffer the chain kernels read (complete up to the sets still open or in flight)
    bool flush_alternate; // EKF_FLUSH_ALTERNATE (default on): dense passes walk the tiles alternately first-to-last and last-to-first
    int flush_dir;        // direction of the next dense pass (0 = first to last)
    // Product tunables (read from the environment at ekf_*_create, listed in include/ekfslam_c.h "Tunables"; every one of them
    // changes scheduling only, never results): EKF_BALANCED_TAIL, EKF_OVERLAP, EKF_PERSIST, EKF_CHAIN_ONE, EKF_CHAIN_HELPERS,
    // EKF_INLINE_REC, EKF_XCD_MAP, ... -- bench.py records every EKF_* variable it saw.
    bool balanced_tail = true;          // EKF_BALANCED_TAIL=0: windows always close at max_pending (launch_ops)
    long long windows_closed = 0;       // windows handed to a dense pass by close_set since create (ekf_debug_windows)
    int last_window_slots = 0;          // ... and the slots of the last one
    // Test / experiment hooks.  They exist only in the debug variant of the library (make debug: -DEKF_DEBUG_HOOKS,
    // libekfslam_hip_debug.so); in the product build the fields keep these values and no environment variable can change them.
    bool dbg_skip_flush = false;        // EKF_DEBUG_SKIP_FLUSH=1: timing experiments only, results are wrong
    int dbg_drop_marks_from = 0;        // EKF_DEBUG_DROP_MARKS_FROM=k (and ..._TO=m, exclusive): dense passes k .. m-1 never report completion (tests of the bounded waits)
    int dbg_drop_marks_to = 0x7fffffff;
    long dbg_stream_idle_ticks = 0;     // EKF_DEBUG_STREAM_IDLE_TICKS=t: a streaming launch leaves by itself after t ticks of the 100 MHz clock without a command (default 10 000)
    bool dbg_stream_no_recheck = false; // EKF_DEBUG_STREAM_NO_RECHECK=1: ... and WITHOUT looking at the command slot once more: commands get lost on purpose (tests of the host's safety net)
    // streaming immediate-mode calls (one-filter handles run by k_chain<true>; EKF_STREAM=0 switches them off): a resident launch consumes
    // the calls' operations from a host-mapped command ring (ekf_device.h: StreamCtl)
    StreamCtl *sctl_h = nullptr;  // host view of dv.sctl
    StreamCtl *sring_h = nullptr; // host view of dv.sring: sctl_h, or the device allocation itself (written through the BAR; never read by the host)
    bool sring_in_hbm = false;
    bool stream_calls = false;    // the handle streams its immediate-mode calls
    bool stream_alive = false;    // a streaming launch has been started and not been told (or seen) to leave
    int stream_launch = 0;        // number of the newest streaming launch (ChainPlan::stream)
    long long stream_starts = 0, stream_ops = 0;  // (diagnostics: ekf_debug_stream)
    long long stream_last_seq = 0;  // sequence number of the newest streamed command
    int stream_slot_before[EKF_STREAM_RING] = {};  // slots of the open window filled BEFORE streamed command seq (index seq % ring): where a replacement launch resumes
    // immediate-mode input ring (host-mapped pinned)
    double *ring_h;
    double *ring_d;
    int ring_ops;  // records in the ring; a record is B*8 doubles
    int ring_pos;
    hipEvent_t ring_ev[2];
    bool ring_ev_valid[2];
    // script
    double *script_d;
    int *cursor_d;
    int script_steps, script_M, script_has_truth;
    std::vector<GraphEntry> graphs;
    // timing
    hipEvent_t t0, t1;
    bool prof_flush;
    std::vector<hipEvent_t> prof_pool;
    size_t prof_used;
    long long prof_launches;
    double prof_ms;
    EkfMirror *mirror_h;  // host view of dv.mirror
    ekf_params params_requested;
    // scratch
    std::vector<int> h_int;
};

static int sticky_status(ekf_batch *h, bool include_capacity);
static int stream_stop(ekf_batch *h);

extern "C" const char *ekf_last_error(void) { return g_last_error.c_str(); }

extern "C" void ekf_default_params(ekf_params *p) {
    if (!p) return;
    p->sigma_v = 0.01;
    p->sigma_w = 0.04;
    p->gamma_max = 50.0;
    p->gamma_min = 10.0;
    p->cond_limit = 80.0;
    p->max_pending = 16;
    p->log_capacity = 4096;
    p->overlap = -1;
}

// EKF_TRACE=1: progress marks of handle creation / destruction on stderr (diagnostic)
static bool trace_on() {
    static int on = -1;
    if (on < 0) on = getenv("EKF_TRACE") ? atoi(getenv("EKF_TRACE")) : 0;
    return on != 0;
}
#define TRACE(msg)                                          \
    do {                                                    \
        if (trace_on()) {                                   \
            fprintf(stderr, "[ekf] %s\n", msg);             \
            fflush(stderr);                                 \
        }                                                   \
    } while (0)

// Streams are recycled through a process-wide pool and never destroyed: hipStreamCreateWithFlags and
// hipExtStreamCreateWithCUMask were seen to block forever once in a few thousand create/destroy cycles (ROCm 7.2;
// scripts/stress_create.py), which a test suite that opens hundreds of handles does reach.  A handle takes an
// idle stream of the right kind (device, CUs kept free: -1 = unmasked) or creates one; ekf_destroy returns it
// after synchronising.
struct PooledStream {
    int device, keep;
    hipStream_t s;
};
static std::mutex g_pool_mu;
static std::vector<PooledStream> g_pool;

static bool pool_take(int device, int keep, hipStream_t *out) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++)
        if (g_pool[i].device == device && g_pool[i].keep == keep) {
            *out = g_pool[i].s;
            g_pool.erase(g_pool.begin() + i);
            return true;
        }
    return false;
}

// Idle streams are destroyed when the process exits, before the HIP runtime's own teardown (atexit handlers run in
// reverse order of registration, and the runtime was loaded first): streams left alive crashed rocprofv3's finalisation.
static void pool_drain() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &p : g_pool) {
        if (hipSetDevice(p.device) == hipSuccess) hipStreamDestroy(p.s);
    }
    g_pool.clear();
}

static void pool_give(int device, int keep, hipStream_t s) {
    static bool registered = false;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!registered) {
        atexit(pool_drain);
        registered = true;
    }
    g_pool.push_back({device, keep, s});
}

// Do kernels on two streams of this process really run side by side?  The overlapped pipeline lets a chain kernel wait
// IN-KERNEL for the dense pass it depends on (cheaper than a cross-stream event by 6 us per window), which is only safe
// when that pass can run while the chain kernel spins.  Tools that serialise kernel execution (rocprofv3 --pmc) break
// the assumption; then the pipeline keeps the event.  Probed once per process and device.
static int concurrent_kernels_ok(int device, hipStream_t a, hipStream_t b) {
    static std::mutex mu;
    static std::vector<int> cache(64, -1);
    std::lock_guard<std::mutex> lk(mu);
    if (device >= 0 && device < (int)cache.size() && cache[device] >= 0) return cache[device];
    int *d = nullptr, h[2] = {0, 0};
    int ok = 0;
    if (hipMalloc((void **)&d, 2 * sizeof(int)) == hipSuccess && hipMemset(d, 0, 2 * sizeof(int)) == hipSuccess) {
        hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(64), 0, a, d, d + 1);
        hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(64), 0, b, d);
        if (hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess &&
            hipMemcpy(h, d, 2 * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess)
            ok = h[1] == 1;
    }
    if (d) hipFree(d);
    (void)hipGetLastError();
    if (device >= 0 && device < (int)cache.size()) cache[device] = ok;
    return ok;
}

template <typename T>
static hipError_t dev_alloc_zero(T **p, size_t count, size_t *total, hipStream_t s) {
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e != hipSuccess) return e;
    *total += bytes;
    return hipMemsetAsync(*p, 0, bytes, s);
}

// Residency registry.  The workgroups of a filter's chain kernel exchange their arg-min candidates while they run, so all
// of them must be resident at once; a workgroup that cannot be placed until another one exits would leave the others
// polling until their bound runs out (EKF_ERR_TIMEOUT).  Every live handle therefore claims the CUs its chain launch needs
// (workgroups / workgroups-per-CU from the occupancy query), and a handle that does not fit beside the live ones is
// refused at creation.  Per process and device; other processes on the GPU are out of sight.
static std::mutex g_res_mu;
static int g_cus_claimed[64];
// CUs the launches of one-workgroup ("solo") handles occupy while they run.  Those handles wait for nobody, so they need no
// co-residency and claim nothing above -- but a 256-filter k_solo launch does fill the GPU, and a multi-segment chain launch of
// ANOTHER handle, whose gated dense passes need free CUs, must know (launch_ops: persist).
static int g_cus_solo[64];

static int create_impl(ekf_batch *h, int batch, int capacity_landmarks, int device_id, const ekf_params *params, const hipDeviceProp_t &prop);

// The debug variant reads its hooks when a handle is created and again at ekf_set_state (a test switches them off between the
// two); the product build compiles to nothing.
static void read_debug_hooks(ekf_batch *h) {
#ifdef EKF_DEBUG_HOOKS
    h->dv.spin_limit = getenv("EKF_DEBUG_SPIN_LIMIT") ? atoll(getenv("EKF_DEBUG_SPIN_LIMIT")) : (1LL << 24);
    h->dbg_skip_flush = getenv("EKF_DEBUG_SKIP_FLUSH") && atoi(getenv("EKF_DEBUG_SKIP_FLUSH")) != 0;
    h->dbg_drop_marks_from = getenv("EKF_DEBUG_DROP_MARKS_FROM") ? atoi(getenv("EKF_DEBUG_DROP_MARKS_FROM")) : 0;
    h->dbg_drop_marks_to = getenv("EKF_DEBUG_DROP_MARKS_TO") ? atoi(getenv("EKF_DEBUG_DROP_MARKS_TO")) : 0x7fffffff;
    h->dbg_stream_idle_ticks = getenv("EKF_DEBUG_STREAM_IDLE_TICKS") ? atol(getenv("EKF_DEBUG_STREAM_IDLE_TICKS")) : 0;
    h->dbg_stream_no_recheck = getenv("EKF_DEBUG_STREAM_NO_RECHECK") && atoi(getenv("EKF_DEBUG_STREAM_NO_RECHECK")) != 0;
#else
    (void)h;
#endif
}

extern "C" int ekf_destroy(ekf_handle h);

extern "C" int ekf_batch_create(ekf_handle *out, int batch, int capacity_landmarks, int device_id, const ekf_params *params) {
    if (!out || batch < 1 || capacity_landmarks < 1 || capacity_landmarks > EKF_MAX_CAPACITY) return set_error(EKF_ERR_BAD_ARG, "bad batch/capacity");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(EKF_ERR_NO_DEVICE, "no HIP device: libekfslam_hip has no CPU fallback");
    if (device_id < 0 || device_id >= ndev || device_id >= 64) return set_error(EKF_ERR_BAD_ARG, "bad device_id");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        char buf[256];
        snprintf(buf, sizeof buf, "device %d is %s; this library carries gfx950 code objects only", device_id, prop.gcnArchName);
        return set_error(EKF_ERR_NO_DEVICE, buf);
    }
    HIP_TRY(hipSetDevice(device_id));
    ekf_batch *h = new ekf_batch();  // value-initialised: every pointer null, every count zero
    h->device = device_id;
    int rc = create_impl(h, batch, capacity_landmarks, device_id, params, prop);
    if (rc != EKF_OK) {
        std::string keep = g_last_error;
        ekf_destroy(h);  // frees whatever was allocated before the failure (tolerates the null members)
        (void)hipGetLastError();
        g_last_error = keep;
        return rc;
    }
    *out = h;
    return EKF_OK;
}

static int create_impl(ekf_batch *h, int batch, int capacity_landmarks, int device_id, const ekf_params *params, const hipDeviceProp_t &prop) {
    ekf_default_params(&h->params);
    if (params) h->params = *params;
    h->params_requested = h->params;  // (ekf_reserve builds the larger handle from what the caller asked for, not from what this capacity allowed)
    if (h->params.max_pending < 1) h->params.max_pending = 1;
    if (h->params.max_pending > EKF_MAX_PENDING) h->params.max_pending = EKF_MAX_PENDING;
    if (h->params.log_capacity < 16) h->params.log_capacity = 16;
    h->device_bytes = 0;
    if (!pool_take(device_id, -1, &h->s_chain)) {
        TRACE("create: stream");
        HIP_TRY(hipStreamCreateWithFlags(&h->s_chain, hipStreamNonBlocking));
        TRACE("create: stream done");
    }

    EkfDev &dv = h->dv;
    memset(&dv, 0, sizeof dv);
    dv.B = batch;
    dv.Ncap = capacity_landmarks;
    dv.T = (2 * capacity_landmarks + 63) / 64;
    dv.xs = ((3 + 64 * dv.T) + 63) / 64 * 64;
    dv.dn = 32 * dv.T;
    dv.logcap = h->params.log_capacity;
    dv.bm_stride = (size_t)dv.T * (dv.T + 1) / 2 * 4096;
    dv.rows = 64 * dv.T;
    dv.gamma_max = h->params.gamma_max;
    dv.gamma_min = h->params.gamma_min;
    dv.spin_limit = 1LL << 24;
    dv.cond_limit = h->params.cond_limit;
    {
        // cond >= L  <=>  q r >= kappa (q^2 + r^2); kappa = 1/2 - 1/(L^2 + 1) (-> 1/2 for L = inf: only q = r is skipped).
        // L <= 1: every finite S is skipped (cond >= 1 always): kappa = 0.  NaN stays NaN: nothing is skipped.
        const long double L = h->params.cond_limit;
        long double kappa = L <= 1.0L ? 0.0L : 0.5L - 1.0L / (L * L + 1.0L);
        dv.cond_k2 = (double)(kappa * kappa);
    }
    // k_chain geometry.  About one landmark per worker thread, at most 32 workgroups per filter, and few
    // enough workgroups in total (<= 256) that all of them are resident at once: the cross-workgroup
    // barrier needs every workgroup of a filter running.  Every workgroup keeps its landmarks' rows of every
    // slot of the open window in LDS (64 bytes per landmark and slot), so landmarks-per-workgroup x window
    // must fit the CU's LDS next to the kernel's static 16 KB: more workgroups first, then a shorter window.
    const int max_workers = EKF_CHAIN_MAX_THREADS - 64;
    const long lds_budget = (long)prop.sharedMemPerBlock - 16384;  // (k_chain's static LDS: 16.2 KB)
    if (lds_budget < 64 * 64) return set_error(EKF_ERR_NO_DEVICE, "device reports too little LDS per workgroup");
    int maxp = h->params.max_pending;
    // overlap (params.overlap, EKF_OVERLAP overrides): automatic = on when two windows of every landmark's slot rows fit
    // the LDS of at most 64 resident workgroups per filter, i.e. when it does not cost window length
    int want_overlap = getenv("EKF_OVERLAP") ? atoi(getenv("EKF_OVERLAP")) : h->params.overlap;
    if (want_overlap < 0) {
        int g_max = batch >= 256 ? 1 : (EKF_CHAIN_MAX_WGS < 256 / batch ? EKF_CHAIN_MAX_WGS : 256 / batch);
        if (g_max < 1) g_max = 1;
        long lpw_min = ((capacity_landmarks + g_max - 1) / g_max + 63) / 64 * 64;  // (the LDS cache is laid out in chunks of 64 landmarks)
        want_overlap = (lpw_min * maxp * 2 * 32 <= lds_budget) ? 1 : 0;
        // ... and when there is a dense pass worth hiding.  Round 4 (scripts/history/r04_geometry.py): with several windows per chain launch
        // the overlapped pipeline also saves the launch boundaries between chain kernel and pass, and wins from P_LL = 10 MB on
        // (N = 768: 39.2 k against 35.3 k steps/s in place; N = 1024: 38.5 k against 35.3 k; N = 2048: 37.8 k against 32.3 k; N = 512,
        // 4 MB: 37.6 k against 36.9 k -- a draw; the threshold is 8 MB).  The threshold was 128 MB in rounds 1-3, measured on one-window launches.
        size_t T = (2 * (size_t)capacity_landmarks + 63) / 64;
        if ((size_t)batch * (T * (T + 1) / 2) * 4096 * sizeof(double) < ((size_t)8 << 20)) want_overlap = 0;
    }
    h->overlap = want_overlap != 0;
    h->params.overlap = h->overlap ? 1 : 0;
    const int sets_in_lds = h->overlap ? 2 : 1;  // overlap: the set being folded by the dense pass in flight is still needed
    int G = (capacity_landmarks + max_workers - 1) / max_workers;
    int G_lds = (int)((((long)capacity_landmarks + 63) / 64 * 64 * maxp * sets_in_lds * 32 + lds_budget - 1) / lds_budget);
    if (G_lds > G) G = G_lds;
    // Round 4: about 64 landmarks -- ONE worker wave -- per workgroup is the fastest shape wherever the GPU has the CUs for it, up to
    // 32 workgroups per filter (fewer waves to keep in step at every barrier; N = 512: 8 workgroups 36.9 k against 3 workgroups
    // 34.4 k steps/s, N = 1024: 16 against 6: 35.3 k against 32.5 k in place, N = 2048: 32 against 16: 37.8 k against 36.3 k
    // overlapped; at N = 4096 the rule gives the 32 workgroups of 128 landmarks the LDS budget asked for already, and 64
    // workgroups of 64 were slower there: 29.7 k against 31.6 k, the dense pass loses too many CUs)
    {
        int G_pref = (capacity_landmarks + 63) / 64;
        if (G_pref > 32) G_pref = 32;
        if (G_pref > G) G = G_pref;
    }
    if (G > EKF_CHAIN_MAX_WGS) G = EKF_CHAIN_MAX_WGS;
    if (G * batch > 256) G = 256 / batch;  // (batches of more than 256 filters: one workgroup per filter, several launches)
    if (G < 1) G = 1;
    // the cache holds whole chunks of 64 landmarks per workgroup: a few more workgroups can save a whole chunk each
    // (N = 4096, window 16, two sets: 28 workgroups of 147 landmarks would need 3 chunks, 32 of 128 need 2)
    {
        auto lds_need = [&](int g) { return ((long)(capacity_landmarks + g - 1) / g + 63) / 64 * 64 * maxp * sets_in_lds * 32; };
        const int g_cap = batch >= 256 ? 1 : (EKF_CHAIN_MAX_WGS < 256 / batch ? EKF_CHAIN_MAX_WGS : 256 / batch);
        while (G < g_cap && lds_need(G) > lds_budget) G++;
    }
    // One workgroup per filter and one slot set ("solo").  Maps of up to 256 landmarks whose window fits one CU's LDS are run by
    // k_solo (ekf_solo.hip): one landmark per thread, no control wave, no exchange, one barrier per measurement.  EKF_SOLO=0
    // keeps k_chain for them (A/B comparisons, tests of k_chain's one-workgroup path).
    const bool want_solo_kernel = !h->overlap && (getenv("EKF_SOLO") ? atoi(getenv("EKF_SOLO")) != 0 : true) && !getenv("EKF_CHAIN_WGS");
    // (k_solo runs windows of up to twice what its cache holds -- ekf_solo.hip, SOLO_HALF -- so 16 slots of cache are enough for any window)
    if (want_solo_kernel && capacity_landmarks <= 256 &&
        ((long)capacity_landmarks + 63) / 64 * 64 * (maxp > 2 * 16 ? maxp : (maxp > 16 ? 16 : maxp)) * 32 <= lds_budget) G = 1;
    if (getenv("EKF_CHAIN_WGS")) G = atoi(getenv("EKF_CHAIN_WGS")) > 0 ? atoi(getenv("EKF_CHAIN_WGS")) : G;
    if (G > EKF_CHAIN_MAX_WGS) G = EKF_CHAIN_MAX_WGS;
    if (G * batch > 256) G = 256 / batch > 0 ? 256 / batch : 1;
    h->solo = !h->overlap && G == 1;
    h->solo_kernel = h->solo && want_solo_kernel && capacity_landmarks <= 256;
    h->stagger_ticks = getenv("EKF_SOLO_STAGGER_US") ? atoi(getenv("EKF_SOLO_STAGGER_US")) * 100 : 3500;
    h->chain_wgs = G;
    h->chain_filters = batch * G <= 256 ? batch : 256 / G;  // every workgroup of a launch resident at once
    dv.gmax = G;
    dv.lpw = (capacity_landmarks + G - 1) / G;
    const long lpw64 = ((long)dv.lpw + 63) / 64 * 64;  // the own-row cache holds whole chunks of 64 landmarks
    int cache_slots = maxp * sets_in_lds;  // slots of own rows in LDS
    if (lpw64 * maxp * sets_in_lds * 32 > lds_budget) {
        const bool two_halves = h->solo_kernel && lds_budget / (lpw64 * 32) >= 16 && !(getenv("EKF_SOLO_LONG_WINDOW") && atoi(getenv("EKF_SOLO_LONG_WINDOW")) == 0);
        if (two_halves) {
            // k_solo: the window's first 16 slots move into registers when the cache is full (ekf_solo.hip: SOLO_HALF): a window of
            // up to 32 with 16 slots of cache -- one dense pass per 32 measurements for a map of 256 landmarks
            if (maxp > 32) maxp = 32;
            cache_slots = 16;
        } else {
            maxp = (int)(lds_budget / (lpw64 * sets_in_lds * 32));
            if (maxp > 1) maxp &= ~1;  // whole slot pairs
            cache_slots = maxp * sets_in_lds;
        }
    }
    if (maxp < 1) return set_error(EKF_ERR_BAD_ARG, "capacity too large for this batch size (one window slot does not fit LDS)");
    h->params.max_pending = maxp;  // the effective window, see ekf_window()
    dv.maxp = maxp;
    dv.maxpairs = (dv.maxp + 1) / 2;
    dv.f_stride = (size_t)(dv.maxpairs + 1) * dv.rows * 4;
    dv.vs_cap = cache_slots;
    h->solo_long = h->solo_kernel && maxp > cache_slots;
    // (the tile of the in-kernel pass lives in a128..a255 -- the registers of a long window's first half, free otherwise --, the A operands of
    // a tile row in the own-row cache, dead while the pass runs: 2 KiB per pair and wave -- 8 or 16 pairs -- against 2 KiB per cached slot)
    h->solo_fuse = h->solo_kernel && cache_slots >= ((((maxp + 1) >> 1) + 7) & ~7) && (getenv("EKF_SOLO_FUSE") ? atoi(getenv("EKF_SOLO_FUSE")) != 0 : true);
    h->chain_lds = (size_t)lpw64 * cache_slots * 32;
    HIP_TRY(hipFuncSetAttribute((const void *)k_chain<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget));  // one setting for every handle
    HIP_TRY(hipFuncSetAttribute((const void *)k_chain<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget));
    HIP_TRY(hipFuncSetAttribute((const void *)k_chain<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget));
    HIP_TRY(hipFuncSetAttribute((const void *)k_chain<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget));
    HIP_TRY(hipFuncSetAttribute((const void *)k_solo<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget));
    HIP_TRY(hipFuncSetAttribute((const void *)k_solo<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget));
    HIP_TRY(hipFuncSetAttribute((const void *)k_solo<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget));
    HIP_TRY(hipFuncSetAttribute((const void *)k_solo<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget));
    int workers = (dv.lpw + 63) / 64 * 64;
    if (workers > max_workers) workers = max_workers;
    if (dv.lpw > 64 && dv.lpw <= 128) workers = 192;  // two owner waves and a third that shares their fold (k_chain: helper_on)
    // one owner wave and two that take a third of its fold each (round 4: N = 768 / 12 workgroups 39.2 k -> 40.5 k steps/s, N = 1024 / 16:
    // 38.0 k -> 39.5 k; with 32 workgroups -- N = 2048 -- the two extra waves at every barrier cost more than the shorter fold gives:
    // 37.0 k -> 35.8 k, so only up to 16 workgroups)
    if (dv.lpw <= 64 && G > 1 && G <= 16) workers = 192;
    // ... and, whatever the number of workgroups, where the fold is long: 48 virtual slots and more (round 5: N = 4096 as 64 workgroups of 64 landmarks
    // with two windows of 32 in LDS: 36.6 k steps/s with the helper waves, 32.9 k without)
    if (dv.lpw <= 64 && G > 1 && cache_slots >= 48) workers = 192;
    if (getenv("EKF_CHAIN_HELPERS") && dv.lpw <= 64 && G > 1) workers = atoi(getenv("EKF_CHAIN_HELPERS")) != 0 ? 192 : (dv.lpw + 63) / 64 * 64;  // (experiments: force / forbid the two helper waves)
    h->chain_threads = 64 + workers;  // wave 0 is the control wave
    if (h->solo_kernel) h->chain_threads = (capacity_landmarks + 63) / 64 * 64;  // k_solo: one landmark per thread, no control wave
    dv.hpw = (dv.lpw + 63) / 64;
    h->chain_one = !h->solo_kernel && G > 1 && dv.lpw <= h->chain_threads - 64 && G * dv.hpw <= EKF_CHAIN_MAX_WGS && !(getenv("EKF_CHAIN_ONE") && atoi(getenv("EKF_CHAIN_ONE")) == 0);
    if (!h->chain_one) dv.hpw = 1;
    dv.nrec = G * dv.hpw;
    {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, h->solo_kernel ? (h->solo_long ? (const void *)k_solo<true> : (const void *)k_solo<false>) : (h->chain_one ? (const void *)k_chain<true> : (const void *)k_chain<false>), h->chain_threads, h->chain_lds));
        if (per_cu < 1) return set_error(EKF_ERR_STATE, "the chain kernel does not fit a CU with this capacity / window");
        // (one-workgroup filters wait for nobody: they need no co-residency and claim nothing)
        const int need = h->solo ? 0 : (G * h->chain_filters + per_cu - 1) / per_cu;
        std::lock_guard<std::mutex> lk(g_res_mu);
        if (g_cus_claimed[device_id] + need > prop.multiProcessorCount) {
            char buf[256];
            snprintf(buf, sizeof buf, "this handle's %d chain workgroups need %d CUs, %d of %d are claimed by live handles: they could not all be resident at once",
                     G * h->chain_filters, need, g_cus_claimed[device_id], prop.multiProcessorCount);
            return set_error(EKF_ERR_STATE, buf);
        }
        g_cus_claimed[device_id] += need;
        h->claimed_cus = need;
        h->solo_cus = h->solo ? (h->chain_filters + per_cu - 1) / per_cu : 0;
        if (h->solo_cus > prop.multiProcessorCount) h->solo_cus = prop.multiProcessorCount;
        g_cus_solo[device_id] += h->solo_cus;
        h->ncu = prop.multiProcessorCount;
    }
    size_t B = batch;
    hipStream_t s = h->s_chain;
    HIP_TRY(dev_alloc_zero(&dv.x, B * dv.xs, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.R, B * 3 * dv.xs, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.D, B * 3 * dv.dn, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.Bm[0], B * dv.bm_stride, &h->device_bytes, s));
    if (h->overlap) {  // the dense pass goes buffer to buffer
        // The second buffer is shifted by 4 KB against the first so that a tile's source and destination differ in DRAM
        // channel/bank phase: 107-108 us per pass instead of 110-111 (scripts/history/exp_skew.sh; EKF_BM_SKEW overrides, bytes)
        size_t skew = (getenv("EKF_BM_SKEW") ? (size_t)atol(getenv("EKF_BM_SKEW")) : (size_t)4096) / sizeof(double);
        HIP_TRY(dev_alloc_zero(&h->bm1_base, B * dv.bm_stride + skew, &h->device_bytes, s));
        dv.Bm[1] = h->bm1_base + skew;
    }
    else dv.Bm[1] = dv.Bm[0];  // one buffer: the dense pass runs in place
    HIP_TRY(dev_alloc_zero(&dv.FA, B * 2 * dv.f_stride, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.FB, B * 2 * dv.f_stride, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.n_lm, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.n_lm_sweep, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.n_lm_flush, B * 2, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.status, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.slot_active, B * 2 * dv.maxp + EKF_MAX_PENDING, &h->device_bytes, s));  // (+ EKF_MAX_PENDING: k_flush_rb reads that many entries of a row unconditionally)
    HIP_TRY(dev_alloc_zero(&dv.slot_meta, B * 2 * dv.maxp, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.pass_flag, 1, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.seg_count, EKF_PLAN_MAX, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.bar, B * 2, &h->device_bytes, s));
#ifdef EKF_CHAIN_STAMPS
    HIP_TRY(dev_alloc_zero(&dv.dbg, 32 + 65 * 2048, &h->device_bytes, s));  // + publish times of every workgroup, 2048 exchanges (scripts/history/r04_skew.py)
#else
    HIP_TRY(dev_alloc_zero(&dv.dbg, 32, &h->device_bytes, s));
#endif
    HIP_TRY(dev_alloc_zero(&dv.part, B * 2 * dv.nrec * EKF_REC_DOUBLES, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.log, B * dv.logcap, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.log_count, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.stats, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&h->cursor_d, 1, &h->device_bytes, s));

    TRACE("create: device buffers queued");
    HIP_TRY(hipHostMalloc((void **)&h->mirror_h, B * sizeof(EkfMirror), hipHostMallocMapped));
    memset(h->mirror_h, 0, B * sizeof(EkfMirror));
    HIP_TRY(hipHostGetDevicePointer((void **)&dv.mirror, h->mirror_h, 0));
    HIP_TRY(dev_alloc_zero(&dv.sfw, 40, &h->device_bytes, s));
    HIP_TRY(hipHostMalloc((void **)&h->sctl_h, sizeof(StreamCtl), hipHostMallocMapped));
    memset(h->sctl_h, 0, sizeof(StreamCtl));
    HIP_TRY(hipHostGetDevicePointer((void **)&dv.sctl, h->sctl_h, 0));
    {
        // the command ring in device memory where the host can write it (large BAR; EKF_STREAM_RING_HOST=1 keeps it in host memory)
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, h->device) != hipSuccess) large_bar = 0, (void)hipGetLastError();
        const bool want_host = getenv("EKF_STREAM_RING_HOST") && atoi(getenv("EKF_STREAM_RING_HOST")) != 0;
        dv.sring = dv.sctl, h->sring_h = h->sctl_h;
        if (large_bar && !want_host) {
            StreamCtl *d = nullptr;
            if (hipExtMallocWithFlags((void **)&d, sizeof(StreamCtl), hipDeviceMallocFinegrained) == hipSuccess) {
                HIP_TRY(hipMemsetAsync(d, 0, sizeof(StreamCtl), s));
                h->device_bytes += sizeof(StreamCtl);
                dv.sring = d, h->sring_h = d, h->sring_in_hbm = true;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    size_t rec_bytes = B * 8 * sizeof(double);
    long ring_ops = (long)((16u << 20) / rec_bytes);
    if (ring_ops > 1024) ring_ops = 1024;
    if (ring_ops < 2 * EKF_CHAIN_MAX_OPS) ring_ops = 2 * EKF_CHAIN_MAX_OPS;
    h->ring_ops = (int)ring_ops;
    HIP_TRY(hipHostMalloc((void **)&h->ring_h, rec_bytes * h->ring_ops, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&h->ring_d, h->ring_h, 0));
    h->ring_pos = 0;
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipEventCreateWithFlags(&h->ring_ev[i], hipEventDisableTiming));
        h->ring_ev_valid[i] = false;
    }
    HIP_TRY(hipEventCreate(&h->t0));
    HIP_TRY(hipEventCreate(&h->t1));
    h->balanced_tail = !(getenv("EKF_BALANCED_TAIL") && atoi(getenv("EKF_BALANCED_TAIL")) == 0);
    h->prof_flush = false;
    h->prof_used = 0;
    h->prof_launches = 0;
    h->prof_ms = 0;
    h->n_lm_hi = 0;
    h->cur_set = 0;
    h->pending = 0;
    h->buf_in = 0;
    h->prev_pending = 0;
    h->ev_idx = 0;
    h->chain_signalled = false;
    h->chain_seq = 0;
    h->pass_seq = 0;
    h->need_pass = 0;
    h->inkernel_wait = false;
    h->mirror_by_chain = false;
    h->s_flush = h->s_chain;
    if (h->overlap) {
        // The chain kernel needs its workgroups' CUs the moment it is launched; a dense pass that owns every CU
        // would make it queue behind whole tiles.  The dense pass therefore gets a stream restricted to the CUs
        // the chain does not need (EKF_CHAIN_CUS overrides the number kept free).
        int keep = getenv("EKF_CHAIN_CUS") ? atoi(getenv("EKF_CHAIN_CUS")) : (G * batch < 32 ? G * batch : 32);  // (measured: 32 beats 64 even for 64 workgroups)
        int ncu = prop.multiProcessorCount;
        if (keep > ncu / 2) keep = ncu / 2;
        h->flush_keep = keep > 0 ? keep : -1;
        if (!pool_take(device_id, h->flush_keep, &h->s_flush)) {
            hipError_t em = hipErrorUnknown;
            if (keep > 0) {
                std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
                // CU i sits on XCD i % 8 (round-robin numbering): free the same share of every XCD
                int per_xcd = (keep + 7) / 8, xcds = 8, freed[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < ncu; i++) {
                    int xc = i % xcds;
                    bool chain_cu = freed[xc] < per_xcd;
                    if (chain_cu) freed[xc]++;
                    else mask[i / 32] |= 1u << (i % 32);
                }
                TRACE("create: cu-mask stream");
                em = hipExtStreamCreateWithCUMask(&h->s_flush, (uint32_t)mask.size(), mask.data());
                TRACE("create: cu-mask stream done");
            }
            if (em != hipSuccess) {
                (void)hipGetLastError();
                h->flush_keep = -1;  // an ordinary stream (and that is the pool it goes back to)
                if (!pool_take(device_id, -1, &h->s_flush)) HIP_TRY(hipStreamCreateWithFlags(&h->s_flush, hipStreamNonBlocking));
            }
        }
        h->flush_masked = h->flush_keep > 0;
        // In-kernel waiting only where the pass has a queue of its own CUs and kernels of the two streams were seen to run
        // side by side; everywhere else (unmasked fallback stream, serialising tools) the chain stream waits for the
        // pass's event.
        h->inkernel_wait = (getenv("EKF_INKERNEL_WAIT") ? atoi(getenv("EKF_INKERNEL_WAIT")) != 0 : true) && h->flush_masked &&
                           concurrent_kernels_ok(device_id, h->s_chain, h->s_flush) != 0;
        HIP_TRY(hipEventCreate(&h->ev_chain));  // (stop events of dispatch packets)
        for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreate(&h->ev_flush[i]));
        h->chain_signalled = false;
    }
    if (h->solo) {
        int ng = getenv("EKF_SOLO_GROUPS") ? atoi(getenv("EKF_SOLO_GROUPS")) : 1;
        if (ng > 8) ng = 8;
        if (ng < 1 || batch < 16 * ng) ng = 1;  // (small batches: one launch for all filters)
        h->ngroups = ng;
        h->s_grp.assign(ng, nullptr);
        h->s_grp[0] = h->s_chain;
        h->ev_join.assign(ng, nullptr);
        if (ng > 1) HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int g = 1; g < ng; g++) {
            if (!pool_take(device_id, -1, &h->s_grp[g])) HIP_TRY(hipStreamCreateWithFlags(&h->s_grp[g], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_join[g], hipEventDisableTiming));
        }
    }
    h->flush_alternate = getenv("EKF_FLUSH_ALTERNATE") ? atoi(getenv("EKF_FLUSH_ALTERNATE")) != 0 : true;
    {
        int can_wait_value = 0;
        (void)hipDeviceGetAttribute(&can_wait_value, hipDeviceAttributeCanUseStreamWaitValue, device_id);
        h->persist = (getenv("EKF_PERSIST") ? atoi(getenv("EKF_PERSIST")) != 0 : true) && can_wait_value != 0;
    }
    h->flush_dir = 0;
    read_debug_hooks(h);
    // streaming immediate-mode calls: every handle of ONE filter (k_chain above 256 landmarks, k_solo up to 256)
    h->stream_calls = batch == 1 && !(getenv("EKF_STREAM") && atoi(getenv("EKF_STREAM")) == 0);
    h->xcd_map = getenv("EKF_XCD_MAP") ? atoi(getenv("EKF_XCD_MAP")) != 0 : true;
    h->batch_interleave = getenv("EKF_BATCH_INTERLEAVE") ? atoi(getenv("EKF_BATCH_INTERLEAVE")) != 0 : true;
    h->script_d = nullptr;
    h->script_steps = h->script_M = h->script_has_truth = 0;
    h->h_int.resize(B);
    TRACE("create: final sync");
    HIP_TRY(hipStreamSynchronize(h->s_chain));
    TRACE("create: done");
    return EKF_OK;
}

extern "C" int ekf_create(ekf_handle *out, int capacity_landmarks, int device_id, const ekf_params *params) {
    return ekf_batch_create(out, 1, capacity_landmarks, device_id, params);
}

// Bounds-checking variant (make check: -DEKF_CHAIN_CHECK, libekfslam_hip_check.so): every data-dependent global index of k_chain
// is range-checked on the device and the first violation of a handle is left in dv.dbg[8..11]; a handle that is destroyed with one
// reports it on stderr and counts it here (tests/test_safety_builds.py runs part of the suite against that library and wants 0).
static std::atomic<long> g_check_violations{0};
extern "C" long ekf_debug_check_violations(void) { return g_check_violations.load(); }

extern "C" int ekf_destroy(ekf_handle h) {
    if (!h) return EKF_OK;
    hipSetDevice(h->device);
    TRACE("destroy: sync");
    if (h->sctl_h && h->s_chain) (void)stream_stop(h);  // (a resident streaming launch leaves first)
    if (h->s_chain) hipStreamSynchronize(h->s_chain);
#ifdef EKF_CHAIN_CHECK
    if (h->dv.dbg) {
        long long d[12] = {0};
        if (hipMemcpy(d, h->dv.dbg, sizeof d, hipMemcpyDeviceToHost) == hipSuccess && d[8] != 0) {
            g_check_violations++;
            fprintf(stderr, "EKF_CHAIN_CHECK: index %lld outside [0, %lld) at ekf_kernels.hip:%lld (workgroup * 100000 + thread = %lld)\n", d[9], d[10], d[8], d[11]);
        }
    }
#endif
    if (h->s_flush && h->s_flush != h->s_chain) {
        hipStreamSynchronize(h->s_flush);
        pool_give(h->device, h->flush_keep, h->s_flush);
    }
    for (size_t g = 1; g < h->s_grp.size(); g++)
        if (h->s_grp[g]) {
            hipStreamSynchronize(h->s_grp[g]);
            pool_give(h->device, -1, h->s_grp[g]);
        }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    for (auto e : h->ev_join)
        if (e) hipEventDestroy(e);
    if (h->ev_chain) hipEventDestroy(h->ev_chain);
    for (int i = 0; i < 2; i++)
        if (h->ev_flush[i]) hipEventDestroy(h->ev_flush[i]);
    if (h->bm1_base) hipFree(h->bm1_base);
    for (auto &g : h->graphs) hipGraphExecDestroy(g.exec);
    EkfDev &dv = h->dv;
    hipFree(dv.x), hipFree(dv.R), hipFree(dv.D), hipFree(dv.Bm[0]), hipFree(dv.FA), hipFree(dv.FB);
    hipFree(dv.n_lm), hipFree(dv.n_lm_sweep), hipFree(dv.n_lm_flush), hipFree(dv.status), hipFree(dv.slot_active), hipFree(dv.slot_meta), hipFree(dv.pass_flag), hipFree(dv.seg_count);
    hipFree(dv.bar), hipFree(dv.part), hipFree(dv.dbg);
    hipFree(dv.log), hipFree(dv.log_count), hipFree(dv.stats);
    hipFree(h->cursor_d);
    for (int *m : h->tile_maps)
        if (m) hipFree(m);
    if (h->script_d) hipFree(h->script_d);
    if (h->ring_h) hipHostFree(h->ring_h);
    if (h->sring_in_hbm) hipFree(h->dv.sring);
    if (h->sctl_h) hipHostFree(h->sctl_h);
    if (h->dv.sfw) hipFree(h->dv.sfw);
    if (h->mirror_h) hipHostFree(h->mirror_h);
    for (int i = 0; i < 2; i++)
        if (h->ring_ev[i]) hipEventDestroy(h->ring_ev[i]);
    if (h->t0) hipEventDestroy(h->t0);
    if (h->t1) hipEventDestroy(h->t1);
    for (auto e : h->prof_pool) hipEventDestroy(e);
    TRACE("destroy: streams");
    if (h->s_chain) pool_give(h->device, -1, h->s_chain);
    if (h->claimed_cus > 0 || h->solo_cus > 0) {
        std::lock_guard<std::mutex> lk(g_res_mu);
        g_cus_claimed[h->device] -= h->claimed_cus;
        g_cus_solo[h->device] -= h->solo_cus;
    }
    delete h;
    (void)hipGetLastError();
    TRACE("destroy: done");
    return EKF_OK;
}

extern "C" int ekf_batch_size(ekf_handle h) { return h ? h->dv.B : EKF_ERR_BAD_ARG; }
extern "C" int ekf_window(ekf_handle h) { return h ? h->dv.maxp : EKF_ERR_BAD_ARG; }
extern "C" int ekf_overlap(ekf_handle h) { return h ? (h->overlap ? 1 : 0) : EKF_ERR_BAD_ARG; }

extern "C" int ekf_capacity(ekf_handle h) { return h ? h->dv.Ncap : EKF_ERR_BAD_ARG; }
extern "C" void *ekf_stream(ekf_handle h) {
    if (!h) return nullptr;
    (void)stream_stop(h);  // (the caller is about to order its own work on this stream: nothing resident may hold it)
    return (void *)h->s_chain;
}
extern "C" size_t ekf_device_bytes(ekf_handle h) { return h ? h->device_bytes : 0; }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Low-latency waits: a blocking hipStreamSynchronize / hipEventSynchronize costs 20-50 us of wake-up latency, which is a
// sizeable part of a run of a few hundred microseconds.  The host polls the stream / event for up to two milliseconds
// (about a microsecond per query) and only then blocks.
// ... and a blocking wait in the runtime is never entered at all: once in a few dozen waits of several milliseconds the wake-up
// came 4-10 ms late on the gpurun boxes (scripts/history/r04_stall_hunt.py: hipEventSynchronize returned 10-16 ms after the start of a
// 6.9 ms region; the driver's box showed 54 ms once), which is what a robot loop must not see.  So: spin on the query for 2 ms
// (short waits: no system call), then keep querying between naps (one core mostly asleep, wake-up bounded by the nap).
// The naps back off with the length of the wait -- 20 us up to 10 ms (with the kernel's default 50 us timer slack a nap really lasts
// about 70 us), 100 us up to 100 ms, 1 ms beyond: with eight ranks on a host and waits of many milliseconds (ekf_sync at the end of a
// timed region, ekf_reserve) a rank wakes its host thread a few thousand times per second at most instead of tens of thousands, a
// wait of a second costs about a thousand queries, and a hung GPU is polled a thousand times per second, not spun on.
template <typename Query>
static hipError_t poll_wait(Query query) {
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long spin = 0;; spin++) {
        hipError_t e = query();
        if (e != hipErrorNotReady) return e;
        if ((spin & 63) == 63) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 2000000L) break;
        }
        __builtin_ia32_pause();
    }
    for (;;) {
        hipError_t e = query();
        if (e != hipErrorNotReady) return e;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const long waited_us = (t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000L;
        const timespec nap = {0, waited_us < 10000L ? 20000L : (waited_us < 100000L ? 100000L : 1000000L)};
        nanosleep(&nap, nullptr);
    }
}
static hipError_t stream_wait(hipStream_t s) {
    return poll_wait([s]() { return hipStreamQuery(s); });
}

static hipError_t event_wait(hipEvent_t ev) {
    return poll_wait([ev]() { return hipEventQuery(ev); });
}

static int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof buf, "kernel launch failed: %s", hipGetErrorString(e));
        return set_error(EKF_ERR_HIP, buf);
    }
    return EKF_OK;
}

// Wave -> tile table of the row-block dense pass for nT live tile rows: workgroup w (4 waves) gets tiles of class
// (I mod 2, J mod 4) = w mod 8, consecutive tiles of a class sharing their tile row; a class that runs dry takes from the
// fullest one.  Built once per nT (the map only grows), uploaded synchronously.
static const int *tile_map_for(ekf_batch *h, int nT) {
    if (!h->xcd_map || nT < 8) return nullptr;  // small maps: nothing to share
    if ((int)h->tile_maps.size() <= nT) h->tile_maps.resize(nT + 1, nullptr);
    if (h->tile_maps[nT]) return h->tile_maps[nT];
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->s_chain, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;  // no allocation or copy while a graph is being captured: this pass uses the arithmetic order
    }
    std::vector<std::vector<int>> cls(8);
    for (int I = 0; I < nT; I++)
        for (int J = I; J < nT; J++) cls[(I & 1) * 4 + (J & 3)].push_back((I << 16) | J);
    std::vector<size_t> pos(8, 0);
    const int total = nT * (nT + 1) / 2, nwg = (total + 3) / 4;
    std::vector<int> map((size_t)nwg * 4, -1);
    for (int w = 0; w < nwg; w++)
        for (int k = 0; k < 4; k++) {
            int c = w & 7;
            if (pos[c] >= cls[c].size()) {  // this class is used up: take from the class with most tiles left
                size_t best = 0;
                c = -1;
                for (int q = 0; q < 8; q++)
                    if (cls[q].size() - pos[q] > best) best = cls[q].size() - pos[q], c = q;
                if (c < 0) break;
            }
            map[(size_t)w * 4 + k] = cls[c][pos[c]++];
        }
    int *d = nullptr;
    if (hipMalloc((void **)&d, map.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(d);
        return nullptr;
    }
    h->device_bytes += map.size() * sizeof(int);
    h->tile_maps[nT] = d;
    return d;
}

// ---- chain / dense-pass alternation ------------------------------------------------------------------
// Close the slot set being filled and hand it to a dense pass; the chain continues into the other set.
//  * without overlap: the pass folds the set into Bm in place, in stream order behind the chain kernels.
//  * with overlap: pass k runs on its own stream, Bm[fin] -> Bm[fin ^ 1], while the chain kernels of window
//    k+1 read Bm[fin] and fold set k themselves (n_prev).  Pass k starts after the chain kernels of window k
//    (ev_chain) and, by stream order, after pass k-1 whose output it reads; the chain kernels of window k+1
//    start after pass k-1 (they read its output and overwrite the slot rows it read).
typedef std::vector<std::function<hipError_t()>> EnqueueList;

// The profiling events (a start / stop pair per dense pass, or per k_solo launch that folds its own windows) are created where
// profiling is switched on and where a script is loaded -- enough pairs for every window the script can close between two reads --
// never on the enqueue path: four hipEventCreate calls inside a timed region of 0.85 ms were part of what the round-5 driver
// run paid.  The enqueue path still grows the pool when a caller outruns it (immediate-mode traffic without a read).
static int prof_reserve(ekf_batch *h, size_t pairs) {
    if (pairs > 4096) pairs = 4096;
    while (h->prof_pool.size() < h->prof_used + 2 * pairs) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->prof_pool.push_back(e);
    }
    return EKF_OK;
}
static size_t prof_pairs_for_script(const ekf_batch *h) {
    if (!h->script_d) return 8;
    // a window closes after at least maxp / 2 measurements (the balanced tail) -- and once more per call (terminal passes)
    const size_t slots = (size_t)h->script_steps * (size_t)(h->script_M > 0 ? h->script_M : 0);
    const size_t half = (size_t)(h->dv.maxp > 1 ? h->dv.maxp / 2 : 1);
    return (slots / half + 8) * (size_t)(h->ngroups > 1 ? h->ngroups : 1);
}

// ---- streaming immediate-mode calls (ekf_device.h: StreamCtl; k_chain<ONE, true>) -------------------------------------------------
// Start a streaming launch on the open window: one segment without operations, consuming commands from consumed0 + 1 on.
static int stream_start(ekf_batch *h, long long consumed0, int slot0) {
    ChainPlan plan;
    memset(&plan, 0, sizeof plan);
    ChainSeg &sg = plan.s[0];
    sg.k0 = 0, sg.nops = 0, sg.slot0 = slot0, sg.set = h->cur_set, sg.buf_read = h->buf_in, sg.n_prev = h->prev_pending;
    sg.need_pass = h->need_pass, sg.drop = 0;
    sg.seq = consumed0;  // the last command consumed before this launch
    // (a one-workgroup filter whose launches fold the windows they fill does so in a streaming launch too, behind the command that closes the window)
    sg.gate = 0, sg.self_pass = (h->solo_kernel && h->solo_fuse && !h->dbg_skip_flush) ? 1 : 0, sg.stagger = 0;
    plan.nseg = 1, plan.signal = 0, plan.inl_n = 0;
    if (h->dbg_stream_idle_ticks > 0 || h->dbg_stream_no_recheck) {  // (debug library only: a streaming launch has no inline record, the fields carry the test hooks)
        plan.inl_n = 1 | (h->dbg_stream_no_recheck ? 2 : 0);
        plan.inl[0] = h->dbg_stream_idle_ticks > 0 ? (double)h->dbg_stream_idle_ticks : (double)EKF_STREAM_IDLE_TICKS;
    }
    h->stream_launch = h->stream_launch >= 0xffff ? 1 : h->stream_launch + 1;  // (16 bits of it tag the forwards)
    plan.stream = h->stream_launch;
    __atomic_store_n(&h->sctl_h->state, ((unsigned long long)(unsigned)plan.stream << 2) | EKF_STREAM_RUNNING, __ATOMIC_SEQ_CST);
    if (h->solo_kernel && h->solo_long)
        hipExtLaunchKernelGGL((k_solo<true, true>), dim3(1, 1), dim3(h->chain_threads), h->chain_lds, h->s_chain, nullptr, nullptr, 0, h->dv,
                              (const double *)h->ring_d, (const int *)nullptr, plan, 0);
    else if (h->solo_kernel)
        hipExtLaunchKernelGGL((k_solo<false, true>), dim3(1, 1), dim3(h->chain_threads), h->chain_lds, h->s_chain, nullptr, nullptr, 0, h->dv,
                              (const double *)h->ring_d, (const int *)nullptr, plan, 0);
    else if (h->chain_one)
        hipExtLaunchKernelGGL((k_chain<true, true>), dim3(h->chain_wgs, 1), dim3(h->chain_threads), h->chain_lds, h->s_chain, nullptr, nullptr, 0, h->dv,
                              (const double *)h->ring_d, (const int *)nullptr, plan, 0);
    else
        hipExtLaunchKernelGGL((k_chain<false, true>), dim3(h->chain_wgs, 1), dim3(h->chain_threads), h->chain_lds, h->s_chain, nullptr, nullptr, 0, h->dv,
                              (const double *)h->ring_d, (const int *)nullptr, plan, 0);
    h->stream_alive = true;
    h->chain_signalled = false;
    h->stream_starts++;
    return check_launch();
}

// Wait until the streamed command `seq` has been executed (the mirror's sequence number reaches it).  This is also the safety net under
// the leave-by-itself handshake: a launch that has LEFT without consuming what was posted (state EXITED, consumed < seq) is replaced by a
// new one that starts behind what the old one did consume -- the commands are still in the ring -- so that no posted command can be lost
// whatever the order in which the two sides' accesses to host memory were served.  Bounded.
static int stream_wait_consumed(ekf_batch *h, long long seq) {
    StreamCtl *c = h->sctl_h;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int relaunches = 0;
    for (long spin = 0;; spin++) {
        if (__atomic_load_n(&h->mirror_h[0].seq, __ATOMIC_ACQUIRE) >= seq) return EKF_OK;
        const unsigned long long launch = (unsigned long long)(unsigned)h->stream_launch;
        if (__atomic_load_n(&c->state, __ATOMIC_ACQUIRE) == ((launch << 2) | EKF_STREAM_EXITED)) {
            const long long consumed = (long long)__atomic_load_n(&c->consumed, __ATOMIC_ACQUIRE);
            if (__atomic_load_n(&h->mirror_h[0].seq, __ATOMIC_ACQUIRE) >= seq) return EKF_OK;
            if (h->mirror_h[0].status == EKF_ERR_TIMEOUT) return sticky_status(h, false);  // the launch gave up: the state is invalid, say so
            if (consumed >= seq || ++relaunches > 4) return set_error(EKF_ERR_STATE, "a streaming launch left without publishing what it consumed");
            // (queued behind the old launch; it resumes at the slot the first unconsumed command was posted at -- the window has not changed:
            // a window only closes behind a command that is known to have been executed)
            int rc = stream_start(h, consumed, h->stream_slot_before[(unsigned long long)(consumed + 1) % EKF_STREAM_RING]);
            if (rc) return rc;
        }
        if ((spin & 255) == 255) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 4000000000L) return set_error(EKF_ERR_TIMEOUT, "a streamed command was not executed within 4 s");
        }
        __builtin_ia32_pause();
    }
}

// The resident launch is told to leave and waited for (its epilogue writes the state back that later launches and copies read).  Every
// entry point that enqueues on the chain stream, reads device memory or hands the stream out comes through here first; the
// immediate-mode operations themselves and the mirror's readers (pose, landmark count, robot block, newest decisions, counters) do not.
static int stream_stop(ekf_batch *h) {
    if (!h->stream_alive) return EKF_OK;
    // what has been posted is executed first (the launch looks at the command slot before it looks at the stop word, but two reads of
    // host memory may be served in either order: a stop seen without the command posted in front of it would leave that command behind)
    int rc = stream_wait_consumed(h, h->stream_last_seq);
    h->stream_alive = false;
    // (a plain store: the word may live in device memory behind the BAR, where a sequentially consistent store -- an xchg on x86 -- would be a
    // locked read-modify-write across PCIe; the mapping is write-combining: the fence sends it out now)
    __atomic_store_n(&h->sring_h->stop, (unsigned long long)(unsigned)h->stream_launch, __ATOMIC_RELAXED);
    __builtin_ia32_sfence();
    HIP_TRY(stream_wait(h->s_chain));
    return rc;
}

static int close_set(ekf_batch *h, bool terminal, EnqueueList *defer);

// One immediate-mode operation through the streaming launch: post the command, make sure somebody consumes it, keep the host's
// window bookkeeping (launch_ops' for a one-operation launch).  rec: the operation's record; consumes: it takes a slot.
// n_slots: slots of the open window the command consumes (an operation: 0 or 1; a scripted chunk: its measurements -- the caller cuts
// chunks so that none passes the end of the window).
static int stream_op(ekf_batch *h, const double *rec, int n_slots) {
    StreamCtl *c = h->sctl_h;
    const bool consumes = n_slots > 0;
    const bool closes = consumes && h->pending + n_slots >= h->dv.maxp;
    const long long seq = ++h->chain_seq;
    StreamCmd *cmd = &h->sring_h->cmd[(unsigned long long)seq % EKF_STREAM_RING];
    if (h->stream_alive && seq - EKF_STREAM_RING > 0) {  // the slot's previous command (seq - ring) must have been executed: a caller that posts without ever reading
        int rc = stream_wait_consumed(h, seq - EKF_STREAM_RING);
        if (rc) {
            h->chain_seq--;
            return rc;
        }
    }
    {
        // seventeen granules {32 payload bits, the sequence number's low half}; the flags (g[0]) last: the launch polls them, and re-reads
        // every other granule until it carries the tag
        const unsigned long long tg = ((unsigned long long)seq & 0xffffffffull) << 32;
        for (int i = 0; i < 8; i++) {
            unsigned long long bits;
            memcpy(&bits, rec + i, sizeof bits);
            __atomic_store_n(&cmd->g[1 + 2 * i], (bits & 0xffffffffull) | tg, __ATOMIC_RELAXED);
            __atomic_store_n(&cmd->g[2 + 2 * i], (bits >> 32) | tg, __ATOMIC_RELAXED);
        }
        __atomic_store_n(&cmd->g[0], (unsigned long long)(closes ? EKF_STREAM_END_AFTER : 0) | tg, __ATOMIC_RELEASE);
        __builtin_ia32_sfence();  // (a ring in device memory is written through a write-combining mapping: the command leaves the write buffers now)
    }
    h->stream_ops++;
    h->stream_last_seq = seq;
    h->stream_slot_before[(unsigned long long)seq % EKF_STREAM_RING] = h->pending;
    if (!h->stream_alive) {
        // (every earlier streamed command has been executed: a launch is only written off behind stream_wait_consumed -- stream_stop, a closing command)
        int rc = stream_start(h, seq - 1, h->pending);
        if (rc) return rc;
    } else {
        // "write mine, fence, read yours": the launch does the same with its state word and this command slot before it leaves by itself, so
        // normally one of the two sees the other.  RUNNING: the launch will find the command.  Anything else: wait for the outcome (the
        // operation done, or the launch gone without it and replaced).
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        const unsigned long long launch = (unsigned long long)(unsigned)h->stream_launch;
        if (__atomic_load_n(&c->state, __ATOMIC_ACQUIRE) != ((launch << 2) | EKF_STREAM_RUNNING)) {
            int rc = stream_wait_consumed(h, seq);
            if (rc) return rc;
        }
    }
    h->mirror_by_chain = true;
    h->stats_in_mirror = true;
    if (consumes) h->pending += n_slots;
    if (closes) {
        // the launch leaves behind this operation by itself (EKF_STREAM_END_AFTER); the window's dense pass is enqueued behind it once the
        // closing command is known to have been executed (a synchronising caller would wait for it next anyway)
        int rc = stream_wait_consumed(h, seq);
        h->stream_alive = false;
        if (rc) return rc;
        // The next window's streaming launch goes out at once, IN FRONT of the pass's launch: a pass that is already streaming holds every CU
        // its mask allows with a queue of workgroups behind them, and a chain launch that arrives later gets its 64 CUs one by one as tiles
        // finish -- the first exchange then waits for the last workgroup, 100 us (seen as a bimodal p90).  Launched first, it is resident when
        // the next call comes (within the idle time) and the pass takes what is left.  The event the pass waits for sits between the two.
        if (h->overlap) {
            HIP_TRY(hipEventRecord(h->ev_chain, h->s_chain));  // (behind the launch that filled the window)
            h->chain_signalled = true;
            EnqueueList pass;
            rc = close_set(h, false, &pass);  // the host's state now describes the next window; the pass's calls wait in `pass`
            if (rc) return rc;
            rc = stream_start(h, seq, h->pending);
            for (auto &enq : pass) {
                hipError_t e = enq();
                if (e != hipSuccess && rc == EKF_OK) rc = set_error(EKF_ERR_HIP, hipGetErrorString(e));
            }
            return rc;
        }
        if (h->solo_kernel && h->solo_fuse && !h->dbg_skip_flush) {
            // (folded by the launch itself, behind the closing command: the next launch starts an empty window, as launch_ops has it)
            h->windows_closed++;
            h->last_window_slots = h->pending;
            h->cur_set ^= 1;
            h->pending = 0;
            return stream_start(h, seq, 0);
        }
        rc = close_set(h, false, nullptr);  // (in place: the pass runs on the chain's stream, the next launch behind it)
        if (rc) return rc;
        return stream_start(h, seq, h->pending);
    }
    return EKF_OK;
}

static int close_set(ekf_batch *h, bool terminal = false, EnqueueList *defer = nullptr) {
    if (h->pending == 0) return stream_stop(h);
    {
        int rc_s = stream_stop(h);
        if (rc_s) return rc_s;
    }
    int nT_hi = (2 * h->n_lm_hi + 63) / 64;
    const int fin = (h->overlap && h->prev_pending > 0) ? h->buf_in ^ 1 : h->buf_in;
    // terminal: the caller asked for everything to be folded (ekf_flush, settle), so no chain kernel will run beside this
    // pass.  It then goes out on the chain's own stream (no cross-stream event hop: ~16 us, no k_mark: ~9 us), on all CUs,
    // and in place (the faster form when nothing else uses the memory system); the pipeline restarts empty afterwards.
    terminal = terminal && h->overlap;
    const int fout = (h->overlap && !terminal) ? fin ^ 1 : fin;
    hipStream_t sf = terminal ? h->s_chain : h->s_flush, sc = h->s_chain;
    // What the pass waits for.  A set filled by a segment of a multi-segment chain launch: the value dv.seg_count shows when
    // every workgroup has finished that segment (a stream gate, hipStreamWaitValue64: the kernel is still running).  Otherwise
    // the chain launch's event.
    const unsigned long long gate_seq = terminal ? 0 : h->open_set_gate;
    const int gate_idx = h->open_gate_idx;
    h->open_set_gate = 0;
    bool record_chain = false, wait_chain = false;
    hipEvent_t wait_prev = nullptr;
    if (terminal) {
        if (h->prev_pending > 0) wait_prev = h->pass_done[h->ev_idx];  // pass k-1 wrote Bm[fin]
        h->chain_signalled = false;
    } else if (h->overlap) {
        if (!gate_seq) record_chain = !h->chain_signalled, wait_chain = true;
        h->chain_signalled = false;
    }
    const bool do_pass = !h->dbg_skip_flush && (nT_hi > 0 || h->overlap);
    if (nT_hi < 1) nT_hi = 1;
    const int total = nT_hi * (nT_hi + 1) / 2;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (do_pass) {
        if (h->overlap && !terminal) e1 = h->ev_flush[h->ev_idx ^ 1];  // pass k's completion, signalled by its own dispatch packet
        if (h->prof_flush) {
            if (h->prof_pool.size() < h->prof_used + 2) {  // (a caller that outran the pool of ekf_flush_profile / ekf_script_load)
                int rc = prof_reserve(h, 8);
                if (rc) return rc;
            }
            e0 = h->prof_pool[h->prof_used++], e1 = h->prof_pool[h->prof_used++];  // (recycled after a read, which leaves both streams idle; e1 doubles as the pass's completion event)
        }
    }
    // Passes alternate direction: a pass starts on the tiles the previous pass touched last, which are the ones the
    // 256 MB Infinity Cache still holds (P_LL of N=4096 is 270 MB per buffer: walked the same way every time, the
    // cache has evicted a tile long before the next pass comes back to it).
    const int rev = h->flush_alternate ? h->flush_dir : 0;
    if (do_pass) h->flush_dir ^= 1;
    const int nwg = cdiv(total, 4);
    const bool interleave = h->dv.B > 1 && h->batch_interleave;  // a filter's workgroups on one XCD
    const int *tmap = (do_pass && !interleave && h->dv.B == 1) ? tile_map_for(h, nT_hi) : (const int *)nullptr;
    const EkfDev dv = h->dv;
    const int set = h->cur_set, nslots = h->pending, B = h->dv.B;
    h->windows_closed++;
    h->last_window_slots = nslots;
    bool mark = false, record_done = false, wait_done_on_chain = false, serial = false;
    int mark_value = 0;
    hipEvent_t done_ev = nullptr;
    if (terminal) {
        h->need_pass = 0;
        h->buf_in = fin;
        h->prev_pending = 0;
    } else if (h->overlap) {
        // the chain kernels of the next window depend on pass k-1 (they read its output and overwrite the slot rows it
        // read): they wait for its number in dv.pass_flag themselves
        // ... or, where kernels of two streams do not run side by side, the stream waits for the pass's event
        if (h->inkernel_wait) {
            h->need_pass = h->prev_pending > 0 ? h->pass_seq : 0;  // pass_seq still names pass k-1 here
        } else {
            h->need_pass = 0;
            if (h->prev_pending > 0) wait_prev = h->pass_done[h->ev_idx];  // pass k-1, awaited by the chain's stream
        }
        mark = true, mark_value = ++h->pass_seq;  // pass k
        if (h->dbg_drop_marks_from > 0 && mark_value >= h->dbg_drop_marks_from && mark_value < h->dbg_drop_marks_to) mark = false;  // (test hook of the debug variant: a pass that never reports)
        h->ev_idx ^= 1;  // pass_done[ev_idx] is pass k's completion: ev_flush[ev_idx] as its stop event, or the profiling pair's stop event
        record_done = false;
        done_ev = (h->prof_flush && do_pass) ? e1 : h->ev_flush[h->ev_idx];
        if (!do_pass) record_done = true, done_ev = h->ev_flush[h->ev_idx];  // (no pass kernel to carry the event: a marker)
        h->pass_done[h->ev_idx] = done_ev;
        serial = getenv("EKF_OVERLAP_SERIAL") != nullptr;  // experiment: no concurrency
        wait_done_on_chain = serial;
        h->buf_in = fin;
        h->prev_pending = h->pending;
    }
    h->cur_set ^= 1;
    h->pending = 0;
    hipEvent_t ev_chain = h->ev_chain;
    auto enqueue = [=]() -> hipError_t {
        hipError_t e = hipSuccess;
        if (terminal) {
            if (wait_prev && (e = hipStreamWaitEvent(sc, wait_prev, 0)) != hipSuccess) return e;
        } else {
            if (gate_seq && (e = hipStreamWaitValue64(sf, dv.seg_count + gate_idx, gate_seq, hipStreamWaitValueGte, 0xffffffffffffffffull)) != hipSuccess) return e;
            if (record_chain && (e = hipEventRecord(ev_chain, sc)) != hipSuccess) return e;
            if (wait_chain && (e = hipStreamWaitEvent(sf, ev_chain, 0)) != hipSuccess) return e;
        }
        if (do_pass) {
            // (start/stop events ride on the dispatch packet itself: no extra barrier packets)
            if (interleave) {
                dim3 g1((unsigned)(cdiv(B, 8) * 8 * nwg), 1);
                hipExtLaunchKernelGGL(k_flush_rb, g1, dim3(256), 0, sf, e0, e1, 0, dv, nT_hi, set, nslots, fin, fout, (const int *)nullptr, nwg, rev, 0, B);
            } else {
                hipExtLaunchKernelGGL(k_flush_rb, dim3(nwg, B), dim3(256), 0, sf, e0, e1, 0, dv, nT_hi, set, nslots, fin, fout, tmap, 0, rev, 0, B);
            }
        }
        if (!terminal && wait_prev && (e = hipStreamWaitEvent(sc, wait_prev, 0)) != hipSuccess) return e;
        if (mark) hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, sf, dv.pass_flag, mark_value);
        if (record_done && (e = hipEventRecord(done_ev, sf)) != hipSuccess) return e;
        if (wait_done_on_chain && (e = hipStreamWaitEvent(sc, done_ev, 0)) != hipSuccess) return e;
        return hipGetLastError();
    };
    if (defer) {
        defer->push_back(enqueue);
        return EKF_OK;
    }
    hipError_t e = enqueue();
    if (e != hipSuccess) return set_error(EKF_ERR_HIP, hipGetErrorString(e));
    return EKF_OK;
}

// Everything folded into Bm[buf_in], streams idle.
static int settle(ekf_batch *h) {
    int rc = close_set(h, true);
    if (rc) return rc;
    HIP_TRY(stream_wait(h->s_chain));
    if (h->overlap) {
        HIP_TRY(stream_wait(h->s_flush));
        if (h->prev_pending > 0) {
            h->buf_in ^= 1;  // the last pass's output
            h->prev_pending = 0;
        }
    }
    return EKF_OK;
}

// Launch k_chain over ops [k0, k0 + nops) of `in`, cutting at slot-set boundaries.
// consumes[i] != 0 when op i takes a slot (measurement, masked measurement, compass).
// defer_last_close: when the last launch of this call fills the set, leave the set closed-to-be: the next call closes it
// (regular pass) or ekf_flush / a state read does (terminal pass, close_set).  Scripted runs use it; never while capturing.
static int launch_ops_grouped(ekf_batch *h, const double *in, int k0, const unsigned char *consumes, int nops);

static int launch_ops(ekf_batch *h, const double *in, const int *cursor, int k0, const unsigned char *consumes, int nops, bool defer_last_close = false) {
    if (h->stream_calls && cursor == nullptr && in == h->ring_d && nops > 0) {
        // an immediate-mode call of a one-filter handle: its operations go to the resident streaming launch, one command each
        if (h->pending == h->dv.maxp) {  // a set whose close a scripted run deferred: more work follows, regular pass
            int rc = close_set(h);
            if (rc) return rc;
        }
        for (int q = 0; q < nops; q++) {
            int rc = stream_op(h, h->ring_h + (size_t)(k0 + q) * 8, consumes[q] ? 1 : 0);
            if (rc) return rc;
        }
        return EKF_OK;
    }
    if (h->stream_calls && cursor == nullptr && in == h->script_d && nops > 0 && nops <= EKF_CHAIN_MAX_OPS) {
        // A SHORT scripted chunk of a one-filter handle (ekf_script_run of a step or two: the per-step call pattern of BASELINE.json config 2's
        // latency figure): one OP_SCRIPT command to the resident launch per stretch that stays inside the open window -- at most HALF a window
        // of measurements in all: a step or two.  Whole windows and longer runs keep the multi-segment launches (their windows turn over
        // without the launch leaving, a window they fill exactly stays open for the next call -- what bench.py's timed regions and its
        // one-window `alone` runs rely on).
        int slots = 0;
        for (int q = 0; q < nops; q++) slots += consumes[q] ? 1 : 0;
        if (2 * slots <= h->dv.maxp) {
            if (h->pending == h->dv.maxp) {
                int rc = close_set(h);
                if (rc) return rc;
            }
            int q0 = 0;
            while (q0 < nops) {
                int q1 = q0, used = 0;
                while (q1 < nops && !(consumes[q1] && h->pending + used == h->dv.maxp)) used += consumes[q1] ? 1 : 0, q1++;  // (up to, and including, the measurement that fills the window)
                double rec[8] = {0, 0, 0, 0, 0, 0, 0, (double)OP_SCRIPT};
                const long long bits = (long long)(size_t)h->script_d;
                memcpy(&rec[0], &bits, sizeof bits);
                rec[1] = (double)(k0 + q0), rec[2] = (double)(q1 - q0);
                int rc = stream_op(h, rec, used);
                if (rc) return rc;
                q0 = q1;
            }
            return EKF_OK;
        }
    }
    {
        int rc = stream_stop(h);
        if (rc) return rc;
    }
    int i = 0;
    if (h->pending == h->dv.maxp && nops > 0) {  // a set whose close the previous call deferred: more work follows, regular pass
        int rc = close_set(h);
        if (rc) return rc;
    }
    // Scripted runs in overlap mode: the launches of one call become the segments of one launch (up to EKF_PLAN_MAX at a
    // time).  The workgroups stay resident across window boundaries -- no launch gap (5 us), no refill of the LDS caches
    // from memory (7 us per 16-measurement window at N = 4096) -- and the dense passes are enqueued behind stream gates that
    // the running kernel opens (dv.seg_count).  Needs kernels of two streams side by side (the in-kernel pass wait's probe).
    // ... and CUs on which a pass can run while the chain workgroups stay resident: a chain wave leaves too few registers
    // on its SIMD for a pass wave, so a pass only runs on CUs without a chain workgroup.  Multi-segment launches are used
    // while the chain workgroups of ALL live handles hold at most half of the GPU (256 filters of one workgroup each would
    // wait forever for a pass that cannot start: that batch keeps one launch per segment).
    bool persist = h->persist && defer_last_close && cursor == nullptr && h->overlap && h->inkernel_wait && h->chain_filters == h->dv.B;
    if (persist) {
        std::lock_guard<std::mutex> lk(g_res_mu);
        persist = (g_cus_claimed[h->device] + g_cus_solo[h->device] - h->solo_cus) * 2 <= h->ncu;  // (everybody else's chain workgroups, solo launches included)
    }
    const bool solo = h->solo_kernel;  // one-workgroup filters run by k_solo
    // k_solo folds a window it fills itself (ChainSeg::self_pass): the launches of one call are the segments of one launch, with
    // no dense-pass launch between them (EKF_SOLO_FUSE=0: one launch per window and k_flush_rb, for comparisons)
    const bool fuse = solo && h->solo_fuse;
    const bool multi = persist || fuse;
    if (!fuse && h->solo && h->ngroups > 1 && cursor == nullptr) {
        long slots = h->pending;
        for (int q = 0; q < nops; q++) slots += consumes[q] ? 1 : 0;
        if (slots >= 2L * h->dv.maxp) return launch_ops_grouped(h, in, k0, consumes, nops);  // the call closes at least two windows (scripted runs, and immediate-mode chunks that long)
    }
    ChainPlan plan;
    memset(&plan, 0, sizeof plan);
    EnqueueList passes;
    int next_drop = 0;
    static const bool inline_rec = !(getenv("EKF_INLINE_REC") && atoi(getenv("EKF_INLINE_REC")) == 0);
    auto launch_plan = [&](hipEvent_t stop_ev) -> int {
        if (plan.nseg == 0) return EKF_OK;
        // one filter, one segment of one operation, its record in the host-mapped ring (an immediate-mode call, slam.cpp:136-170): the
        // record rides in the kernel arguments -- the kernel's first trip to host memory brings it along, the ring costs a second one
        plan.inl_n = 0;
        if (inline_rec && h->dv.B == 1 && plan.nseg == 1 && plan.s[0].nops == 1 && cursor == nullptr && in == h->ring_d) {
            memcpy(plan.inl, h->ring_h + (size_t)plan.s[0].k0 * 8, 8 * sizeof(double));
            plan.inl_n = 1;
        }
        hipEvent_t pe0 = nullptr, pe1 = nullptr;
        if (fuse) {
            int sp = 0;
            for (int q = 0; q < plan.nseg; q++) sp += plan.s[q].self_pass;
            // EKF_SOLO_FUSE_STAGGER_US: a one-off phase shift between the filters of a batch (four groups), so that their own passes take
            // turns in HBM.  Measured: nothing to gain (6.53 M filter-steps/s without, 6.51 / 6.42 / 6.35 M with 35 / 67 / 90 us: one wave
            // per SIMD is latency-bound on its own tile, not bandwidth-bound), so the default is none.
            static const int fuse_stagger = getenv("EKF_SOLO_FUSE_STAGGER_US") ? atoi(getenv("EKF_SOLO_FUSE_STAGGER_US")) * 100 : 0;
            plan.s[0].stagger = (sp >= 4 && h->dv.B >= 16) ? fuse_stagger : 0;
            if (h->prof_flush && sp > 0) {  // the passes live inside this launch: time the launch, count the passes
                if (h->prof_pool.size() < h->prof_used + 2) {
                    int rc = prof_reserve(h, 8);
                    if (rc) return rc;
                }
                pe0 = h->prof_pool[h->prof_used++], pe1 = h->prof_pool[h->prof_used++];
                h->prof_fused_passes += sp;
                h->prof_solo_pairs++;
            }
        }
        for (int b0 = 0; b0 < h->dv.B; b0 += h->chain_filters) {
            const int nb = h->dv.B - b0 < h->chain_filters ? h->dv.B - b0 : h->chain_filters;
            const bool last = b0 + nb >= h->dv.B;
            if (solo) {
                hipEvent_t ev0 = b0 == 0 ? pe0 : nullptr, ev1 = last ? (pe1 ? pe1 : stop_ev) : nullptr;
                if (h->solo_long) hipExtLaunchKernelGGL(k_solo<true>, dim3(1, nb), dim3(h->chain_threads), h->chain_lds, h->s_chain, ev0, ev1, 0, h->dv, in, cursor, plan, b0);
                else hipExtLaunchKernelGGL(k_solo<false>, dim3(1, nb), dim3(h->chain_threads), h->chain_lds, h->s_chain, ev0, ev1, 0, h->dv, in, cursor, plan, b0);
            } else if (h->chain_one)
                hipExtLaunchKernelGGL(k_chain<true>, dim3(h->chain_wgs, nb), dim3(h->chain_threads), h->chain_lds, h->s_chain, nullptr, last ? stop_ev : nullptr, 0, h->dv, in,
                                      cursor, plan, b0);
            else
                hipExtLaunchKernelGGL(k_chain<false>, dim3(h->chain_wgs, nb), dim3(h->chain_threads), h->chain_lds, h->s_chain, nullptr, last ? stop_ev : nullptr, 0, h->dv, in,
                                      cursor, plan, b0);
        }
        if (plan.signal)
            for (int q = 0; q < plan.nseg; q++) h->seg_count_base[q] = plan.s[q].gate;
        plan.nseg = 0;
        next_drop = 0;
        for (auto &enq : passes) {
            hipError_t e = enq();
            if (e != hipSuccess) {
                passes.clear();
                return set_error(EKF_ERR_HIP, hipGetErrorString(e));
            }
        }
        passes.clear();
        return check_launch();
    };
    // Balanced tail (round 5; multi-workgroup filters in the overlapped multi-segment mode, EKF_BALANCED_TAIL=0 switches it off): when between one
    // and two windows' worth of measurements are left in this call, the window that begins is closed at HALF of them (rounded up to a whole
    // slot pair) instead of at max_pending.  A scripted run of 20 steps x 4 measurements at a window of 32 ran 32 | 32 | 16: the pass of the
    // second window (16 pairs, 165 us beside the chain kernel) outlasted the 16 measurements after it (100 us) and the last window's pass
    // waited behind it -- 170 us exposed after the chain kernel's end; 32 | 24 | 24 hides the second pass under the third segment and leaves
    // one 12-pair pass exposed.  Nothing changes for a run whose length is a multiple of the window, nor in the steady state of a long one.
    const bool balanced_tail = h->balanced_tail;
    int slots_left = 0;  // slot-consuming operations of this call from op i on
    for (int q = 0; q < nops; q++) slots_left += consumes[q] ? 1 : 0;
    int limit = h->dv.maxp;  // where the window being filled closes (a window carried over from an earlier call: max_pending)
    while (i < nops) {
        int start = i, used = h->pending;
        if (used == 0) {
            limit = h->dv.maxp;
            if (balanced_tail && persist && !solo && slots_left > h->dv.maxp && slots_left < 2 * h->dv.maxp) {
                limit = ((slots_left + 1) / 2 + 1) & ~1;
                if (limit > h->dv.maxp) limit = h->dv.maxp;  // (an odd max_pending: rounding up to a slot pair must not pass the window -- slot_meta rows, the own-row cache and the pass are sized for maxp)
            }
        }
        while (i < nops && i - start < EKF_CHAIN_MAX_OPS) {
            if (consumes[i]) {
                if (used == limit) break;
                used++;
                slots_left--;
            }
            i++;
        }
        if (i == start) {  // set already full (cannot happen: full sets are closed above and below)
            int rc = close_set(h, false, persist ? &passes : nullptr);
            if (rc) return rc;
            continue;
        }
        // overlap: the launch that fills the set signals ev_chain from its own dispatch packet (no marker packet)
        const bool closes = h->overlap && used == limit;
        ChainSeg sg;
        sg.k0 = k0 + start, sg.nops = i - start, sg.slot0 = h->pending, sg.set = h->cur_set, sg.buf_read = h->buf_in, sg.n_prev = h->prev_pending;
        if (next_drop > 0 && next_drop < h->prev_pending) {  // (the LDS shift moves whole windows: start a new launch instead)
            int rc = launch_plan(nullptr);
            if (rc) return rc;
        }
        sg.need_pass = h->need_pass, sg.drop = next_drop;
        sg.seq = ++h->chain_seq;  // (one number per segment: every filter's mirror reaches it)
        sg.gate = h->seg_count_base[plan.nseg] + (unsigned long long)h->chain_wgs * h->dv.B;  // every workgroup of the launch has finished this segment
        sg.self_pass = (fuse && used == limit && !h->dbg_skip_flush) ? 1 : 0;
        sg.stagger = 0;
        next_drop = 0;
        plan.s[plan.nseg++] = sg;
        plan.signal = persist ? 1 : 0;
        plan.stream = 0;
        h->mirror_by_chain = true;
        h->stats_in_mirror = true;
        h->pending = used;
        if (!multi) {
            int rc = launch_plan(closes ? h->ev_chain : nullptr);
            if (rc) return rc;
            h->chain_signalled = closes;
        } else {
            h->chain_signalled = false;
            if (persist && used == limit) h->open_set_gate = sg.gate, h->open_gate_idx = plan.nseg - 1;
        }
        if (sg.self_pass) {
            h->cur_set ^= 1;  // folded by the launch itself: the next segment starts an empty window
            h->pending = 0;
        } else if (used == limit && !(defer_last_close && i == nops)) {
            int rc = close_set(h, false, persist ? &passes : nullptr);
            if (rc) return rc;
            next_drop = sg.n_prev;  // the next segment starts a window: the set whose pass has finished leaves the LDS caches
        }
        if (multi && plan.nseg == EKF_PLAN_MAX) {
            int rc = launch_plan(nullptr);
            if (rc) return rc;
        }
    }
    return launch_plan(nullptr);
}

// One-workgroup filters in phase groups: a call that closes several windows (a scripted run) is cut by filter range instead of
// running the whole batch in lock step.  Group g's stream carries chain launch, dense pass (in place), chain launch, ... for its
// filters only; nothing orders the groups against each other (filters are independent), and a one-off delay in front of each
// group's first launch spreads their phases, so that the passes take turns in HBM while the other groups' chain kernels -- which
// are latency-bound and leave the memory system idle -- run.  Fork from and join into s_chain, so that every other entry point
// keeps seeing one stream.
static int launch_ops_grouped(ekf_batch *h, const double *in, int k0, const unsigned char *consumes, int nops) {
    const int ng = h->ngroups, B = h->dv.B, maxp = h->dv.maxp;
    const int per = (B + ng - 1) / ng;
    hipError_t e;
    if ((e = hipEventRecord(h->ev_fork, h->s_chain)) != hipSuccess) return set_error(EKF_ERR_HIP, hipGetErrorString(e));
    for (int g = 1; g < ng; g++) {
        if ((e = hipStreamWaitEvent(h->s_grp[g], h->ev_fork, 0)) != hipSuccess) return set_error(EKF_ERR_HIP, hipGetErrorString(e));  // (nothing has been launched on a group stream yet)
        if (h->stagger_ticks > 0) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, h->s_grp[g], (long long)g * h->stagger_ticks);
    }
    const bool interleave = h->batch_interleave;
    // The group streams have been forked off s_chain: whatever happens from here on, they are joined back before this function
    // returns -- later work on s_chain (ring reuse, memsets of ekf_set_state, the hipFree of a staging buffer) must not run beside
    // chain and pass kernels still queued on a group stream.  An error on the way is remembered, the join still happens (where even
    // the join's event calls fail the group stream is drained on the host).
    int rc_pending = EKF_OK;
    if (h->prof_flush) {  // (event creation can fail: do it in front of the first launch, not between a fork and its join)
        long passes = 0, used_ = h->pending;
        for (int q = 0; q < nops; q++)
            if (consumes[q] && ++used_ == maxp) passes++, used_ = 0;
        while (h->prof_pool.size() < h->prof_used + 2 * (size_t)ng * (size_t)(passes + 1)) {
            hipEvent_t ev;
            if (hipEventCreate(&ev) != hipSuccess) {
                rc_pending = set_error(EKF_ERR_HIP, "hipEventCreate failed (dense-pass profiling)");
                break;
            }
            h->prof_pool.push_back(ev);
        }
    }
    int i = 0;
    while (i < nops && rc_pending == EKF_OK) {
        int start = i, used = h->pending;
        while (i < nops && i - start < EKF_CHAIN_MAX_OPS) {
            if (consumes[i]) {
                if (used == maxp) break;
                used++;
            }
            i++;
        }
        ChainPlan plan;
        memset(&plan, 0, sizeof plan);
        ChainSeg &sg = plan.s[0];
        sg.k0 = k0 + start, sg.nops = i - start, sg.slot0 = h->pending, sg.set = h->cur_set, sg.buf_read = h->buf_in;
        sg.seq = ++h->chain_seq;
        sg.gate = 0, sg.self_pass = 0, sg.stagger = 0;
        plan.nseg = 1;
        h->mirror_by_chain = true;
        h->stats_in_mirror = true;
        h->pending = used;
        const bool fold = used == maxp;
        int nT_hi = (2 * h->n_lm_hi + 63) / 64;
        const bool do_pass = fold && !h->dbg_skip_flush && nT_hi > 0;
        const int total = nT_hi * (nT_hi + 1) / 2, nwg = cdiv(total, 4);
        const int rev = h->flush_alternate ? h->flush_dir : 0;
        for (int g = 0; g < ng; g++) {
            const int b0 = g * per, nb = B - b0 < per ? B - b0 : per;
            if (nb <= 0) break;
            if (h->solo_kernel && h->solo_long) hipLaunchKernelGGL(k_solo<true>, dim3(1, nb), dim3(h->chain_threads), h->chain_lds, h->s_grp[g], h->dv, in, (const int *)nullptr, plan, b0);
            else if (h->solo_kernel) hipLaunchKernelGGL(k_solo<false>, dim3(1, nb), dim3(h->chain_threads), h->chain_lds, h->s_grp[g], h->dv, in, (const int *)nullptr, plan, b0);
            else hipLaunchKernelGGL(k_chain<false>, dim3(1, nb), dim3(h->chain_threads), h->chain_lds, h->s_grp[g], h->dv, in, (const int *)nullptr, plan, b0);
            if (!do_pass) continue;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (h->prof_flush && h->prof_used + 2 <= h->prof_pool.size()) e0 = h->prof_pool[h->prof_used++], e1 = h->prof_pool[h->prof_used++];  // (created above)
            if (interleave)
                hipExtLaunchKernelGGL(k_flush_rb, dim3((unsigned)(cdiv(nb, 8) * 8 * nwg), 1), dim3(256), 0, h->s_grp[g], e0, e1, 0, h->dv, nT_hi, h->cur_set, maxp, h->buf_in, h->buf_in,
                                      (const int *)nullptr, nwg, rev, b0, nb);
            else
                hipExtLaunchKernelGGL(k_flush_rb, dim3(nwg, nb), dim3(256), 0, h->s_grp[g], e0, e1, 0, h->dv, nT_hi, h->cur_set, maxp, h->buf_in, h->buf_in, (const int *)nullptr, 0, rev, b0, nb);
        }
        if (fold) {
            if (do_pass) h->flush_dir ^= 1;
            h->cur_set ^= 1;
            h->pending = 0;
        }
    }
    for (int g = 1; g < ng; g++) {
        if ((e = hipEventRecord(h->ev_join[g], h->s_grp[g])) == hipSuccess) e = hipStreamWaitEvent(h->s_chain, h->ev_join[g], 0);
        if (e != hipSuccess) {
            (void)stream_wait(h->s_grp[g]);  // no event to order the streams with: drain this one here
            if (rc_pending == EKF_OK) rc_pending = set_error(EKF_ERR_HIP, hipGetErrorString(e));
        }
    }
    if (rc_pending != EKF_OK) return rc_pending;
    return check_launch();
}

static void bump_bound(ekf_batch *h, int measurements) {
    h->n_lm_hi += measurements;
    if (h->n_lm_hi > h->dv.Ncap) h->n_lm_hi = h->dv.Ncap;
}

// ---- input ring ---------------------------------------------------------------------------------
// Reserve `count` consecutive records (never straddling the wrap); *rec points at the first.
static int ring_reserve(ekf_batch *h, int count, double **rec, int *k_out) {
    int half = h->ring_ops / 2;
    if (count > half) return set_error(EKF_ERR_BAD_ARG, "too many operations in one call");
    int pos = h->ring_pos;
    int which = pos < half ? 0 : 1;
    if (h->stream_calls && pos + count > (which + 1) * half) {
        // (a streaming handle's records are copied into the command ring when they are posted: no launch ever reads this ring)
        which ^= 1;
        pos = which * half;
    } else if (pos + count > (which + 1) * half) {  // does not fit the rest of this half: move to the next half
        HIP_TRY(hipEventRecord(h->ring_ev[which], h->s_chain));
        h->ring_ev_valid[which] = true;
        which ^= 1;
        pos = which * half;
        if (h->ring_ev_valid[which]) HIP_TRY(event_wait(h->ring_ev[which]));  // its readers from one lap ago
    }
    *rec = h->ring_h + (size_t)pos * h->dv.B * 8;
    *k_out = pos;
    h->ring_pos = pos + count;
    return EKF_OK;
}

// Wait until the newest chain launch has written the host-mapped mirror, then read it.  When that launch is the
// newest writer the host spins on the mirror's sequence number (stored last by the kernel, system scope) for up to
// a millisecond -- a stream synchronise costs 10-20 us of wake-up latency per call, which is most of an
// immediate-mode call -- and falls back to the synchronise.  full = true always synchronises (callers that go on to
// read device memory written by the whole launch).
static int refresh_bounds(ekf_batch *h, bool full = true) {
    bool done = false;
    if (!full && h->mirror_by_chain && h->chain_seq > 0) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (long spin = 0;; spin++) {
            bool all = true;
            for (int b = 0; b < h->dv.B; b++)
                if (__atomic_load_n(&h->mirror_h[b].seq, __ATOMIC_ACQUIRE) < h->chain_seq) {
                    all = false;
                    break;
                }
            if (all) {
                done = true;
                break;
            }
            if ((spin & 255) == 255) {
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 1000000L) break;
            }
            __builtin_ia32_pause();
        }
    }
    if (!done) {
        int rc_ = stream_stop(h);  // (a resident streaming launch leaves first: the stream would not drain before its idle time is over)
        if (rc_) return rc_;
        HIP_TRY(stream_wait(h->s_chain));
    }
    int mx = 0;
    for (int b = 0; b < h->dv.B; b++) {
        h->h_int[b] = h->mirror_h[b].n_lm;
        mx = h->h_int[b] > mx ? h->h_int[b] : mx;
    }
    h->n_lm_hi = mx;
    return sticky_status(h, false);  // a timed-out launch left an invalid state: say so wherever state is handed out
}

// ---- propagate --------------------------------------------------------------------------------------
static void make_Q(const ekf_params &p, double v, double Q[4]) {
    // kalmanfilter.cpp:35-37: Q << sv,0,0,sw; Q = (v*v)*Q*Q  ->  ((v*v)*Q)*Q, column-major out
    double a = (v * v) * p.sigma_v, d = (v * v) * p.sigma_w;
    Q[0] = a * p.sigma_v;
    Q[1] = 0.0;
    Q[2] = 0.0;
    Q[3] = d * p.sigma_w;
}

extern "C" int ekf_batch_propagate_q(ekf_handle h, const double *v, const double *w, const double *Q, const double *dt) {
    if (!h || !v || !w || !Q || !dt) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    double *rec;
    int k;
    int rc = ring_reserve(h, 1, &rec, &k);
    if (rc) return rc;
    for (int b = 0; b < h->dv.B; b++) {
        double *r = rec + (size_t)b * 8;
        r[0] = v[b], r[1] = w[b], r[2] = dt[b];
        r[3] = Q[4 * b], r[4] = Q[4 * b + 1], r[5] = Q[4 * b + 2], r[6] = Q[4 * b + 3];
        r[7] = OP_PROP;
    }
    unsigned char consumes = 0;
    return launch_ops(h, h->ring_d, nullptr, k, &consumes, 1);
}

extern "C" int ekf_batch_propagate(ekf_handle h, const double *v, const double *w, const double *dt) {
    if (!h || !v || !w || !dt) return set_error(EKF_ERR_BAD_ARG, "null argument");
    std::vector<double> Q((size_t)4 * h->dv.B);
    for (int b = 0; b < h->dv.B; b++) make_Q(h->params, v[b], &Q[4 * (size_t)b]);
    return ekf_batch_propagate_q(h, v, w, Q.data(), dt);
}

extern "C" int ekf_propagate_q(ekf_handle h, double v, double w, const double Q[4], double dt) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    return ekf_batch_propagate_q(h, &v, &w, Q, &dt);
}

extern "C" int ekf_propagate(ekf_handle h, double v, double w, double dt) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    return ekf_batch_propagate(h, &v, &w, &dt);
}

// ---- update ---------------------------------------------------------------------------------------
static int fetch_decisions(ekf_batch *h, int n_z, ekf_decision *out);

extern "C" int ekf_batch_update(ekf_handle h, const double *z, const double *R, const unsigned char *valid, int n_z, ekf_decision *decisions_out) {
    if (!h || n_z < 0 || (n_z > 0 && (!z || !R))) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    int B = h->dv.B;
    int half = h->ring_ops / 2;
    std::vector<unsigned char> consumes;
    for (int j0 = 0; j0 < n_z; j0 += half) {  // a chunk larger than half the ring goes out in pieces
        int cnt = n_z - j0 < half ? n_z - j0 : half;
        double *rec;
        int k;
        int rc = ring_reserve(h, cnt, &rec, &k);
        if (rc) return rc;
        for (int jj = 0; jj < cnt; jj++) {
            int j = j0 + jj;
            for (int b = 0; b < B; b++) {
                double *r = rec + ((size_t)jj * B + b) * 8;
                const double *zz = z + ((size_t)b * n_z + j) * 2;
                const double *RR = R + ((size_t)b * n_z + j) * 4;
                r[0] = zz[0], r[1] = zz[1];
                r[2] = RR[0], r[3] = RR[1], r[4] = RR[2], r[5] = RR[3];
                r[6] = (j == n_z - 1) ? 2.0 : 1.0;  // 2 = last measurement of the chunk (Update.cpp:26 quirk)
                r[7] = (!valid || valid[(size_t)b * n_z + j]) ? OP_MEAS : OP_SKIP_SLOT;
            }
        }
        consumes.assign(cnt, 1);
        bump_bound(h, cnt);
        rc = launch_ops(h, h->ring_d, nullptr, k, consumes.data(), cnt);
        if (rc) return rc;
    }
    if (decisions_out) return fetch_decisions(h, n_z, decisions_out);
    return EKF_OK;
}

extern "C" int ekf_update(ekf_handle h, const double *z_chunk, const double *R_chunk, int n_z, ekf_decision *decisions_out) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    return ekf_batch_update(h, z_chunk, R_chunk, nullptr, n_z, decisions_out);
}

extern "C" int ekf_batch_update_compass(ekf_handle h, const double *z, const double *R, const unsigned char *valid) {
    if (!h || !z || !R) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    double *rec;
    int k;
    int rc = ring_reserve(h, 1, &rec, &k);
    if (rc) return rc;
    for (int b = 0; b < h->dv.B; b++) {
        double *r = rec + (size_t)b * 8;
        r[0] = z[b], r[1] = R[b];
        r[2] = r[3] = r[4] = r[5] = r[6] = 0;
        r[7] = (!valid || valid[b]) ? OP_COMPASS : OP_SKIP_SLOT;
    }
    unsigned char consumes = 1;
    return launch_ops(h, h->ring_d, nullptr, k, &consumes, 1);
}

extern "C" int ekf_update_compass(ekf_handle h, double z, double R) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    return ekf_batch_update_compass(h, &z, &R, nullptr);
}

extern "C" int ekf_record_truth(ekf_handle h, const double *truth) {
    if (!h || !truth) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    double *rec;
    int k;
    int rc = ring_reserve(h, 1, &rec, &k);
    if (rc) return rc;
    for (int b = 0; b < h->dv.B; b++) {
        double *r = rec + (size_t)b * 8;
        r[0] = truth[3 * b], r[1] = truth[3 * b + 1], r[2] = truth[3 * b + 2];
        r[3] = r[4] = r[5] = r[6] = 0;
        r[7] = OP_TRUTH;
    }
    unsigned char consumes = 0;
    return launch_ops(h, h->ring_d, nullptr, k, &consumes, 1);
}

// ---- synchronising accessors ------------------------------------------------------------------------
// The sticky per-filter status the kernels leave in the host mirror (valid after a synchronising read).  A timeout means
// the filter's state is invalid, and every accessor that hands out state reports it; a capacity overflow leaves a valid
// state (the landmark was simply not added) and is reported by ekf_sync only.
static int sticky_status(ekf_batch *h, bool include_capacity) {
    for (int b = 0; b < h->dv.B; b++) {
        const int st = h->mirror_h[b].status;
        if (st == EKF_ERR_TIMEOUT) {
            char buf[320];
            snprintf(buf, sizeof buf, "filter %d: a device-side wait ran out (the %d chain workgroups of a filter were not all resident -- another tenant on the GPU? -- "
                                      "or the dense pass a launch depends on did not complete); the filter's state is invalid until ekf_set_state", b, h->chain_wgs);
            return set_error(EKF_ERR_TIMEOUT, buf);
        }
        if (st == EKF_ERR_CAPACITY && include_capacity) {
            char buf[160];
            snprintf(buf, sizeof buf, "filter %d: a New landmark did not fit capacity_landmarks = %d (Update.cpp:152-178 would have grown the state)", b, h->dv.Ncap);
            return set_error(EKF_ERR_CAPACITY, buf);
        }
        if (st != 0 && st != EKF_ERR_CAPACITY) return set_error(st, "a kernel reported an error status");
    }
    return EKF_OK;
}

extern "C" int ekf_sync(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    HIP_TRY(stream_wait(h->s_chain));
    if (h->overlap) HIP_TRY(stream_wait(h->s_flush));
    // Every bounded wait that runs out stores EKF_ERR_TIMEOUT into the host-mapped mirror itself, at once (and nothing but
    // k_set_meta ever writes a zero there): the mirror is complete when the streams are idle -- no device-to-host copy here.
    h->h_int.resize(h->dv.B);
    for (int b = 0; b < h->dv.B; b++) h->h_int[b] = h->mirror_h[b].n_lm;
    return sticky_status(h, true);
}

extern "C" int ekf_flush(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return close_set(h, true);
}

extern "C" int ekf_close_window(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return close_set(h, false);
}

extern "C" int ekf_batch_get_pose(ekf_handle h, double *pose_out) {
    if (!h || !pose_out) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = refresh_bounds(h, false);
    if (rc) return rc;
    for (int b = 0; b < h->dv.B; b++)
        for (int i = 0; i < 3; i++) pose_out[3 * b + i] = h->mirror_h[b].pose[i];
    return EKF_OK;
}

extern "C" int ekf_get_pose(ekf_handle h, double pose_out[3]) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    return ekf_batch_get_pose(h, pose_out);
}

extern "C" int ekf_batch_num_landmarks(ekf_handle h, int *n_out) {
    if (!h || !n_out) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = refresh_bounds(h, false);
    if (rc) return rc;
    for (int b = 0; b < h->dv.B; b++) n_out[b] = h->h_int[b];
    return EKF_OK;
}

extern "C" int ekf_num_landmarks(ekf_handle h) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    int n;
    int rc = ekf_batch_num_landmarks(h, &n);
    return rc ? rc : n;
}

extern "C" int ekf_get_robot_cov(ekf_handle h, double P_RR_out[9]) {
    if (!h || h->dv.B != 1 || !P_RR_out) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    HIP_TRY(hipSetDevice(h->device));
    // The chain kernels leave the robot block in the host-mapped mirror beside the pose (round 4: a device-to-host copy here cost the
    // compat shim's doPropagation 15-25 us of every step).  The copy stays for what does not write the mirror last (graph replays).
    if (h->mirror_by_chain && h->chain_seq > 0) {
        int rc = refresh_bounds(h, false);  // waits until the newest launch has written the mirror; sticky EKF_ERR_TIMEOUT
        if (rc == EKF_ERR_TIMEOUT) return rc;
        if (h->mirror_by_chain) {
            for (int i = 0; i < 9; i++) P_RR_out[i] = h->mirror_h[0].Prr[i];
            return EKF_OK;
        }
    }
    HIP_TRY(hipMemcpy2DAsync(P_RR_out, 3 * sizeof(double), h->dv.R, (size_t)h->dv.xs * sizeof(double), 3 * sizeof(double), 3,
                             hipMemcpyDeviceToHost, h->s_chain));
    HIP_TRY(stream_wait(h->s_chain));
    return EKF_OK;
}

extern "C" int ekf_get_x(ekf_handle h, int index, double *x_out, int n_max) {
    if (!h || index < 0 || index >= h->dv.B || !x_out || n_max < 0) return set_error(EKF_ERR_BAD_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    int rc = refresh_bounds(h);
    if (rc) return rc;
    int n = 3 + 2 * h->h_int[index];
    int cnt = n < n_max ? n : n_max;
    HIP_TRY(hipMemcpy(x_out, h->dv.x + (size_t)index * h->dv.xs, sizeof(double) * cnt, hipMemcpyDeviceToHost));
    return n;
}

static int fetch_decisions(ekf_batch *h, int n_z, ekf_decision *out) {
    // [batch][n_z]: the last entries of every filter's log.  Masked measurements leave no entry, so a
    // filter with fewer real entries gets zeroed records in front.
    int B = h->dv.B;
    int rc = refresh_bounds(h, false);  // synchronises
    if (rc) return rc;
    for (int b = 0; b < B; b++) {
        long long cnt = h->mirror_h[b].log_count;
        long long have = cnt < n_z ? cnt : n_z;
        for (int j = 0; j < n_z - have; j++) {
            ekf_decision *dst = out + (size_t)b * n_z + j;
            dst->decision = 0, dst->matched = 0, dst->mahal = 0;
        }
        for (long long j = 0; j < have; j++) {
            long long idx = cnt - have + j;
            ekf_decision *dst = out + (size_t)b * n_z + (n_z - have + j);
            if (cnt - idx <= EKF_MIRROR_DECISIONS) *dst = h->mirror_h[b].last[idx % EKF_MIRROR_DECISIONS];  // still in the host mirror
            else HIP_TRY(hipMemcpy(dst, h->dv.log + (size_t)b * h->dv.logcap + (idx % h->dv.logcap), sizeof(ekf_decision), hipMemcpyDeviceToHost));
        }
    }
    return EKF_OK;
}

extern "C" int ekf_get_decisions(ekf_handle h, int index, ekf_decision *out, int count) {
    if (!h || !out || index < 0 || index >= h->dv.B || count < 0) return set_error(EKF_ERR_BAD_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    long long cnt;
    HIP_TRY(hipMemcpyAsync(&cnt, h->dv.log_count + index, sizeof cnt, hipMemcpyDeviceToHost, h->s_chain));
    HIP_TRY(stream_wait(h->s_chain));
    long long avail = cnt < h->dv.logcap ? cnt : h->dv.logcap;
    long long n = count < avail ? count : avail;
    std::vector<ekf_decision> ring(h->dv.logcap);
    HIP_TRY(hipMemcpy(ring.data(), h->dv.log + (size_t)index * h->dv.logcap, sizeof(ekf_decision) * h->dv.logcap, hipMemcpyDeviceToHost));
    for (long long j = 0; j < n; j++) out[j] = ring[(size_t)((cnt - n + j) % h->dv.logcap)];
    return (int)n;
}

extern "C" int ekf_get_stats(ekf_handle h, ekf_stats *out) {
    if (!h || !out) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (h->mirror_by_chain && h->stats_in_mirror && h->chain_seq > 0) {
        // the newest chain launch copies every filter's counters into the host-mapped mirror: no device-to-host copy
        // (a copy into pageable memory costs 40-100 us, a sizeable part of a short run)
        int rc = refresh_bounds(h, false);
        if (rc) return rc;
        for (int b = 0; b < h->dv.B; b++) out[b] = h->mirror_h[b].stats;
        return EKF_OK;
    }
    HIP_TRY(hipMemcpyAsync(out, h->dv.stats, sizeof(ekf_stats) * h->dv.B, hipMemcpyDeviceToHost, h->s_chain));
    HIP_TRY(stream_wait(h->s_chain));
    return EKF_OK;
}

extern "C" int ekf_stats_means_device(ekf_handle h, double *out_device) {
    if (!h || !out_device) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, out_device) != hipSuccess || (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)) {
        (void)hipGetLastError();
        return set_error(EKF_ERR_BAD_ARG, "ekf_stats_means_device wants device memory");
    }
    hipLaunchKernelGGL(k_stats_means, dim3(cdiv(h->dv.B, 256)), dim3(256), 0, h->s_chain, (const ekf_stats *)h->dv.stats, h->dv.B, out_device);
    HIP_TRY(stream_wait(h->s_chain));
    return check_launch();
}

extern "C" int ekf_reset_stats(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    HIP_TRY(hipMemsetAsync(h->dv.stats, 0, sizeof(ekf_stats) * h->dv.B, h->s_chain));
    h->stats_in_mirror = false;  // until the next chain launch writes the mirror
    return EKF_OK;
}

// ---- dense state injection / extraction -----------------------------------------------------------
extern "C" int ekf_get_state(ekf_handle h, int index, double *x_out, double *P_out, int ld) {
    if (!h || index < 0 || index >= h->dv.B) return set_error(EKF_ERR_BAD_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    int rc = refresh_bounds(h);
    if (rc) return rc;
    int n = 3 + 2 * h->h_int[index];
    if (!x_out && !P_out) return n;
    if (!x_out || !P_out || ld < n) return set_error(EKF_ERR_BAD_ARG, "bad output buffers");
    rc = settle(h);
    if (rc) return rc;
    double *stage = nullptr;  // transient staging: dense n x n + x
    HIP_TRY(hipMalloc((void **)&stage, ((size_t)n * n + n) * sizeof(double)));
    double *xd = stage + (size_t)n * n;
    hipLaunchKernelGGL(k_export, dim3(cdiv(n, 256), n), dim3(256), 0, h->s_chain, h->dv, index, h->buf_in, xd, stage, n, n);
    hipError_t e = hipMemcpyAsync(x_out, xd, sizeof(double) * n, hipMemcpyDeviceToHost, h->s_chain);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(P_out, (size_t)ld * sizeof(double), stage, (size_t)n * sizeof(double), (size_t)n * sizeof(double), n,
                             hipMemcpyDeviceToHost, h->s_chain);
    if (e == hipSuccess) e = stream_wait(h->s_chain);
    hipFree(stage);
    if (e != hipSuccess) return set_error(EKF_ERR_HIP, hipGetErrorString(e));
    return n;
}

extern "C" int ekf_set_state(ekf_handle h, int index, const double *x, const double *P, int ld, int n) {
    if (!h || index < 0 || index >= h->dv.B || !x || !P || n < 3 || ((n - 3) & 1) || ld < n) return set_error(EKF_ERR_BAD_ARG, "bad argument");
    int N = (n - 3) / 2;
    if (N > h->dv.Ncap) return set_error(EKF_ERR_CAPACITY, "state larger than capacity_landmarks");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    int rc = settle(h);
    if (rc) return rc;
    EkfDev &dv = h->dv;
    hipStream_t s = h->s_chain;
    // Both streams are idle here.  A launch that gave up (EKF_ERR_TIMEOUT) leaves the segment counters ahead of the host's
    // bases (workgroups that had not aborted kept counting, open_gates raised the rest) and possibly a pass that never reported:
    // every later stream gate would open early, every later in-kernel wait would run out again.  Start both from zero / from the
    // host's own count, so that a handle is usable again after ekf_set_state, as the sticky status promises.
    read_debug_hooks(h);
    HIP_TRY(hipMemsetAsync(dv.seg_count, 0, sizeof(unsigned long long) * EKF_PLAN_MAX, s));
    for (int q = 0; q < EKF_PLAN_MAX; q++) h->seg_count_base[q] = 0;
    h->open_set_gate = 0;
    if (h->overlap) hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, s, dv.pass_flag, h->pass_seq);
    double *stage = nullptr;
    HIP_TRY(hipMalloc((void **)&stage, ((size_t)n * n + n) * sizeof(double)));
    double *xd = stage + (size_t)n * n;
    hipError_t e = hipMemcpyAsync(xd, x, sizeof(double) * n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(stage, (size_t)n * sizeof(double), P, (size_t)ld * sizeof(double), (size_t)n * sizeof(double), n,
                             hipMemcpyHostToDevice, s);
    size_t b = index;
    if (e == hipSuccess) e = hipMemsetAsync(dv.x + b * dv.xs, 0, sizeof(double) * dv.xs, s);
    if (e == hipSuccess) e = hipMemsetAsync(dv.R + b * 3 * dv.xs, 0, sizeof(double) * 3 * dv.xs, s);
    if (e == hipSuccess) e = hipMemsetAsync(dv.D + b * 3 * dv.dn, 0, sizeof(double) * 3 * dv.dn, s);
    if (e == hipSuccess) e = hipMemsetAsync(dv.Bm[0] + b * dv.bm_stride, 0, sizeof(double) * dv.bm_stride, s);
    if (e == hipSuccess && h->overlap) e = hipMemsetAsync(dv.Bm[1] + b * dv.bm_stride, 0, sizeof(double) * dv.bm_stride, s);
    if (e == hipSuccess) e = hipMemsetAsync(dv.FA + b * 2 * dv.f_stride, 0, sizeof(double) * 2 * dv.f_stride, s);
    if (e == hipSuccess) e = hipMemsetAsync(dv.FB + b * 2 * dv.f_stride, 0, sizeof(double) * 2 * dv.f_stride, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_import, dim3(cdiv(n, 256), n), dim3(256), 0, s, dv, index, h->buf_in, (const double *)xd, (const double *)stage, n, n);
        hipLaunchKernelGGL(k_set_meta, dim3(1), dim3(64), 0, s, dv, index, N);
        h->mirror_by_chain = false;
        e = stream_wait(s);
    }
    hipFree(stage);
    if (e != hipSuccess) return set_error(EKF_ERR_HIP, hipGetErrorString(e));
    if (N > h->n_lm_hi) h->n_lm_hi = N;
    return refresh_bounds(h);
}

// Grow a handle's landmark capacity (the reference grows x and P with every New landmark, Update.cpp:158-177 /
// kalmanfilter.cpp:78-84, and never fails).  A second set of device buffers of the larger capacity is built, every filter's state
// moves over on the device (k_export into a dense staging matrix, k_import from it: the tile numbering depends on the capacity),
// the counters, the decision log and a loaded script move with it, and the handle keeps its address.
extern "C" int ekf_reserve(ekf_handle h, int capacity_landmarks) {
    if (!h || capacity_landmarks < 1 || capacity_landmarks > EKF_MAX_CAPACITY) return set_error(EKF_ERR_BAD_ARG, "bad handle / capacity (EKF_MAX_CAPACITY)");
    if (capacity_landmarks <= h->dv.Ncap) return EKF_OK;
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    int rc = settle(h);  // every deferred slot folded, both streams idle
    if (rc) return rc;
    rc = refresh_bounds(h);  // h_int[b] = landmarks of filter b; a sticky EKF_ERR_TIMEOUT (invalid state) ends it here
    if (rc == EKF_ERR_TIMEOUT) return rc;
    const int B = h->dv.B;
    std::vector<int> n_lm(h->h_int.begin(), h->h_int.begin() + B);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, h->device));
    // the old buffers are idle from here on and will not launch again: their CUs are free for the new handle's residency check
    const int had_claimed = h->claimed_cus, had_solo = h->solo_cus;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        g_cus_claimed[h->device] -= had_claimed;
        g_cus_solo[h->device] -= had_solo;  // (not counted twice while both sets of buffers are alive)
        h->claimed_cus = 0, h->solo_cus = 0;
    }
    ekf_batch *nh = new ekf_batch();
    nh->device = h->device;
    rc = create_impl(nh, B, capacity_landmarks, h->device, &h->params_requested, prop);
    if (rc != EKF_OK) {
        std::string keep = g_last_error;
        ekf_destroy(nh);
        (void)hipGetLastError();
        g_last_error = keep;
        std::lock_guard<std::mutex> lk(g_res_mu);
        g_cus_claimed[h->device] += had_claimed, g_cus_solo[h->device] += had_solo;
        h->claimed_cus = had_claimed, h->solo_cus = had_solo;
        return rc;
    }
    hipError_t e = hipSuccess;
    int n_max = 3;
    for (int b = 0; b < B; b++) n_max = 3 + 2 * n_lm[b] > n_max ? 3 + 2 * n_lm[b] : n_max;
    double *stage = nullptr;
    e = hipMalloc((void **)&stage, ((size_t)n_max * n_max + n_max) * sizeof(double));
    for (int b = 0; b < B && e == hipSuccess; b++) {
        const int n = 3 + 2 * n_lm[b];
        double *xd = stage + (size_t)n * n;
        hipLaunchKernelGGL(k_export, dim3(cdiv(n, 256), n), dim3(256), 0, h->s_chain, h->dv, b, h->buf_in, xd, stage, n, n);
        e = stream_wait(h->s_chain);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_import, dim3(cdiv(n, 256), n), dim3(256), 0, nh->s_chain, nh->dv, b, nh->buf_in, (const double *)xd, (const double *)stage, n, n);
        e = stream_wait(nh->s_chain);
    }
    if (stage) hipFree(stage);
    // counters, decision log (entries name state indices, which do not depend on the capacity), then the bookkeeping kernel
    // (landmark counts, host mirror: pose, count, log position; it also clears the sticky capacity status -- there is room now)
    const bool same_log = nh->dv.logcap == h->dv.logcap;
    if (e == hipSuccess) e = hipMemcpyAsync(nh->dv.stats, h->dv.stats, sizeof(ekf_stats) * B, hipMemcpyDeviceToDevice, nh->s_chain);
    if (e == hipSuccess && same_log) e = hipMemcpyAsync(nh->dv.log, h->dv.log, sizeof(ekf_decision) * (size_t)B * h->dv.logcap, hipMemcpyDeviceToDevice, nh->s_chain);
    if (e == hipSuccess && same_log) e = hipMemcpyAsync(nh->dv.log_count, h->dv.log_count, sizeof(long long) * B, hipMemcpyDeviceToDevice, nh->s_chain);
    if (e == hipSuccess) {
        for (int b = 0; b < B; b++) hipLaunchKernelGGL(k_set_meta, dim3(1), dim3(64), 0, nh->s_chain, nh->dv, b, n_lm[b]);
        e = stream_wait(nh->s_chain);
    }
    if (e != hipSuccess || check_launch() != EKF_OK) {
        std::string keep = e != hipSuccess ? std::string(hipGetErrorString(e)) : g_last_error;
        ekf_destroy(nh);
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(g_res_mu);
        g_cus_claimed[h->device] += had_claimed, g_cus_solo[h->device] += had_solo;
        h->claimed_cus = had_claimed, h->solo_cus = had_solo;
        return set_error(EKF_ERR_HIP, keep.c_str());
    }
    for (int b = 0; b < B; b++) {  // the host mirror's newest decisions and counters (ekf_get_stats / the compat shim read them there)
        memcpy(nh->mirror_h[b].last, h->mirror_h[b].last, sizeof nh->mirror_h[b].last);
        nh->mirror_h[b].stats = h->mirror_h[b].stats;
    }
    nh->stats_in_mirror = h->stats_in_mirror;
    nh->n_lm_hi = h->n_lm_hi;
    nh->prof_flush = h->prof_flush;
    // What a caller may hold across the growth keeps working: an open timer (events are time stamps: t0 recorded on the old buffers'
    // stream pairs with a t1 recorded later), the dense-pass profile collected so far (event pairs and sums), and the stream
    // ekf_stream() handed out -- both chain streams are idle here, so the two handles simply trade them (and every alias of them).
    std::swap(nh->t0, h->t0);
    std::swap(nh->t1, h->t1);
    nh->prof_pool.swap(h->prof_pool);
    nh->prof_used = h->prof_used, h->prof_used = 0;
    nh->prof_ms = h->prof_ms, nh->prof_launches = h->prof_launches;
    nh->prof_fused_passes = h->prof_fused_passes, nh->prof_solo_pairs = h->prof_solo_pairs;
    nh->windows_closed = h->windows_closed, nh->last_window_slots = h->last_window_slots;  // (ekf_debug_windows counts since create, across growths)
    {
        const hipStream_t s_new = nh->s_chain, s_old = h->s_chain;
        if (nh->s_flush == s_new) nh->s_flush = s_old;
        if (h->s_flush == s_old) h->s_flush = s_new;
        if (!nh->s_grp.empty() && nh->s_grp[0] == s_new) nh->s_grp[0] = s_old;
        if (!h->s_grp.empty() && h->s_grp[0] == s_old) h->s_grp[0] = s_new;
        nh->s_chain = s_old, h->s_chain = s_new;
    }
    // a loaded script is laid out per (operation, filter): independent of the capacity
    std::swap(nh->script_d, h->script_d);
    nh->script_steps = h->script_steps, nh->script_M = h->script_M, nh->script_has_truth = h->script_has_truth;
    h->script_steps = 0;
    std::swap(*h, *nh);  // the caller's handle now owns the larger buffers ...
    ekf_destroy(nh);     // ... and the old ones go
    return refresh_bounds(h);
}

extern "C" int ekf_broadcast_state(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    int rc = settle(h);
    if (rc) return rc;
    EkfDev &dv = h->dv;
    hipStream_t s = h->s_chain;
    for (int b = 1; b < dv.B; b++) {
        HIP_TRY(hipMemcpyAsync(dv.x + (size_t)b * dv.xs, dv.x, sizeof(double) * dv.xs, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dv.R + (size_t)b * 3 * dv.xs, dv.R, sizeof(double) * 3 * dv.xs, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dv.D + (size_t)b * 3 * dv.dn, dv.D, sizeof(double) * 3 * dv.dn, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dv.Bm[h->buf_in] + (size_t)b * dv.bm_stride, dv.Bm[h->buf_in], sizeof(double) * dv.bm_stride, hipMemcpyDeviceToDevice, s));
        if (h->overlap) HIP_TRY(hipMemsetAsync(dv.Bm[h->buf_in ^ 1] + (size_t)b * dv.bm_stride, 0, sizeof(double) * dv.bm_stride, s));  // no stale tiles beyond the copied map
        HIP_TRY(hipMemcpyAsync(dv.FA + (size_t)b * 2 * dv.f_stride, dv.FA, sizeof(double) * 2 * dv.f_stride, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dv.FB + (size_t)b * 2 * dv.f_stride, dv.FB, sizeof(double) * 2 * dv.f_stride, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dv.slot_active + (size_t)b * 2 * dv.maxp, dv.slot_active, sizeof(int) * 2 * dv.maxp, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dv.n_lm + b, dv.n_lm, sizeof(int), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dv.n_lm_sweep + b, dv.n_lm_sweep, sizeof(int), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dv.n_lm_flush + 2 * (size_t)b, dv.n_lm_flush, 2 * sizeof(int), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dv.status + b, dv.status, sizeof(int), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dv.log_count + b, dv.log_count, sizeof(long long), hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(stream_wait(s));
    for (int b = 1; b < dv.B; b++) hipLaunchKernelGGL(k_set_meta, dim3(1), dim3(64), 0, s, dv, b, h->mirror_h[0].n_lm);  // also refreshes the host mirror
    h->mirror_by_chain = false;
    return refresh_bounds(h);
}

// ---- scripts --------------------------------------------------------------------------------------
static inline int ops_per_step(const ekf_batch *h) { return 1 + h->script_M + (h->script_has_truth ? 1 : 0); }

extern "C" int ekf_script_load(ekf_handle h, int steps, int M, const double *ctrl, const double *z, const double *R,
                               const unsigned char *valid, const double *truth) {
    if (!h || steps < 1 || M < 0 || !ctrl || (M > 0 && (!z || !R))) return set_error(EKF_ERR_BAD_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    HIP_TRY(stream_wait(h->s_chain));
    for (auto &g : h->graphs) hipGraphExecDestroy(g.exec);
    h->graphs.clear();
    if (h->script_d) {
        HIP_TRY(hipFree(h->script_d));
        h->script_d = nullptr;
    }
    int B = h->dv.B;
    h->script_steps = steps;
    h->script_M = M;
    h->script_has_truth = truth ? 1 : 0;
    int ops = ops_per_step(h);
    size_t count = (size_t)steps * ops * B * 8;
    std::vector<double> host(count, 0.0);
    for (int s = 0; s < steps; s++) {
        double *base = host.data() + (size_t)s * ops * B * 8;
        for (int b = 0; b < B; b++) {
            double *r = base + (size_t)b * 8;
            const double *c = ctrl + ((size_t)s * B + b) * 3;
            double Q[4];
            make_Q(h->params, c[0], Q);
            r[0] = c[0], r[1] = c[1], r[2] = c[2], r[3] = Q[0], r[4] = Q[1], r[5] = Q[2], r[6] = Q[3];
            r[7] = OP_PROP;
        }
        for (int m = 0; m < M; m++)
            for (int b = 0; b < B; b++) {
                double *r = base + ((size_t)(1 + m) * B + b) * 8;
                const double *zz = z + (((size_t)s * M + m) * B + b) * 2;
                const double *RR = R + (((size_t)s * M + m) * B + b) * 4;
                r[0] = zz[0], r[1] = zz[1], r[2] = RR[0], r[3] = RR[1], r[4] = RR[2], r[5] = RR[3];
                r[6] = 2.0;  // every scripted measurement is its own doUpdate call (slam.cpp:150-171)
                r[7] = (!valid || valid[((size_t)s * M + m) * B + b]) ? OP_MEAS : OP_SKIP_SLOT;
            }
        if (truth)
            for (int b = 0; b < B; b++) {
                double *r = base + ((size_t)(1 + M) * B + b) * 8;
                const double *t = truth + ((size_t)s * B + b) * 3;
                r[0] = t[0], r[1] = t[1], r[2] = t[2];
                r[7] = OP_TRUTH;
            }
    }
    HIP_TRY(hipMalloc((void **)&h->script_d, count * sizeof(double)));
    HIP_TRY(hipMemcpy(h->script_d, host.data(), count * sizeof(double), hipMemcpyHostToDevice));
    if (h->prof_flush) return prof_reserve(h, prof_pairs_for_script(h));
    return EKF_OK;
}

// enqueue scripted steps [s0, s0 + ns): op index = (cursor ? *cursor : 0) + k
static int enqueue_script_steps(ekf_batch *h, const int *cursor, int k_first, int ns) {
    int ops = ops_per_step(h), M = h->script_M;
    std::vector<unsigned char> consumes((size_t)ns * ops, 0);
    for (int q = 0; q < ns; q++)
        for (int m = 0; m < M; m++) consumes[(size_t)q * ops + 1 + m] = 1;
    return launch_ops(h, h->script_d, cursor, k_first, consumes.data(), ns * ops, /*defer_last_close*/ cursor == nullptr && h->overlap);
}

static int graph_block_steps(const ekf_batch *h) {
    // smallest S >= 8 whose slot count is an even multiple of maxp, so that a graph ends with an
    // empty slot set and with cur_set / buf_in back where they started
    int M = h->script_M, maxp = h->dv.maxp;
    if (M == 0) return 8;
    for (int S = 8; S <= 8 + 4 * maxp; S++)
        if ((S * M) % (2 * maxp) == 0) return S;
    return 2 * maxp;
}

extern "C" int ekf_script_run(ekf_handle h, int first_step, int n_steps, int use_graph) {
    if (!h || !h->script_d) return set_error(EKF_ERR_STATE, "no script loaded");
    if (first_step < 0 || n_steps < 0 || first_step + n_steps > h->script_steps) return set_error(EKF_ERR_BAD_ARG, "step range outside the script");
    HIP_TRY(hipSetDevice(h->device));
    // (a resident streaming launch takes a short run itself -- launch_ops, OP_SCRIPT -- and leaves first for everything else)
    int ops = ops_per_step(h);
    int s = first_step, end = first_step + n_steps;
    if (use_graph && !h->overlap) {  // (the two-stream pipeline is not captured: plain launches)
        int S = graph_block_steps(h);
        if (end - s >= S) {
            // a graph starts from the settled state (its predecessor in the stream has fully finished)
            int rc = close_set(h);  // (a resident streaming launch leaves here)
            if (rc) return rc;
            GraphEntry *ge = nullptr;
            for (auto &g : h->graphs)
                if (g.steps == S && g.M == h->script_M && g.has_truth == h->script_has_truth) ge = &g;
            int save_set = h->cur_set, save_buf = h->buf_in;
            if (!ge) {
                bool prof_saved = h->prof_flush;
                int hi_saved = h->n_lm_hi;
                h->prof_flush = false;    // event pairs are not captured into graphs
                h->n_lm_hi = h->dv.Ncap;  // graphs bake grid sizes: size the dense pass for the capacity
                (void)tile_map_for(h, (2 * h->n_lm_hi + 63) / 64);  // built before the capture: the captured passes find it ready
                HIP_TRY(hipStreamBeginCapture(h->s_chain, hipStreamCaptureModeThreadLocal));
                int rc2 = enqueue_script_steps(h, h->cursor_d, 0, S);
                if (rc2 == EKF_OK && h->pending != 0) rc2 = set_error(EKF_ERR_STATE, "graph block does not end on an empty slot set");
                hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, h->s_chain, h->cursor_d, S * ops);
                hipGraph_t graph = nullptr;
                hipError_t e = hipStreamEndCapture(h->s_chain, &graph);
                h->prof_flush = prof_saved;
                h->n_lm_hi = hi_saved;
                if (rc2 == EKF_OK && e != hipSuccess) rc2 = set_error(EKF_ERR_HIP, "graph capture failed");
                if (rc2 == EKF_OK && (h->cur_set != save_set || h->buf_in != save_buf)) rc2 = set_error(EKF_ERR_STATE, "graph block does not restore the ping-pong state");
                GraphEntry g;
                g.steps = S, g.M = h->script_M, g.has_truth = h->script_has_truth;
                g.exec = nullptr;
                if (rc2 == EKF_OK && hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) rc2 = set_error(EKF_ERR_HIP, "hipGraphInstantiate failed");
                if (graph) hipGraphDestroy(graph);
                if (rc2) {
                    (void)hipGetLastError();
                    return rc2;
                }
                h->graphs.push_back(g);
                ge = &h->graphs.back();
            }
            int start_op = s * ops;
            HIP_TRY(hipMemcpyAsync(h->cursor_d, &start_op, sizeof(int), hipMemcpyHostToDevice, h->s_chain));
            HIP_TRY(stream_wait(h->s_chain));  // start_op is a stack variable
            while (end - s >= S) {
                HIP_TRY(hipGraphLaunch(ge->exec, h->s_chain));
                s += S;
            }
            // The captured chain launches carry the sequence numbers of the capture: a replay stores those into the host
            // mirror, so mirror.seq says nothing about which replay has finished.  Readers fall back to a stream synchronise.
            h->mirror_by_chain = false;
            h->pending = 0;
            h->n_lm_hi = h->dv.Ncap;  // landmarks may have been appended inside the graphs; unknown until a sync
        }
    }
    if (s < end) {
        bump_bound(h, (end - s) * h->script_M);
        int rc = enqueue_script_steps(h, nullptr, s * ops, end - s);
        if (rc) return rc;
    }
    return check_launch();
}

// Diagnostic tick counters of EKF_CHAIN_STAMPS builds (not declared in the public header).
// (tests) windows closed since create and the slot count of the last one: how launch_ops cut a scripted run
extern "C" int ekf_debug_windows(ekf_handle h, long long *closed, int *last_slots) {
    if (!h || !closed || !last_slots) return set_error(EKF_ERR_BAD_ARG, "null argument");
    *closed = h->windows_closed;
    *last_slots = h->last_window_slots;
    return EKF_OK;
}

// (tests, bench) streaming launches started and operations posted to them since create
extern "C" int ekf_debug_stream(ekf_handle h, long long *starts, long long *ops) {
    if (!h || !starts || !ops) return set_error(EKF_ERR_BAD_ARG, "null argument");
    *starts = h->stream_starts;
    *ops = h->stream_ops;
    return h->stream_calls ? 1 : 0;
}

extern "C" int ekf_debug_stream_ring(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null argument");
    return h->sring_in_hbm ? 1 : 0;
}

extern "C" int ekf_debug_stamps(ekf_handle h, long long *out16, int reset) {
    if (!h || !out16) return EKF_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    HIP_TRY(stream_wait(h->s_chain));
    HIP_TRY(hipMemcpy(out16, h->dv.dbg, 32 * sizeof(long long), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(h->dv.dbg, 0, 32 * sizeof(long long)));
    return EKF_OK;
}

#ifdef EKF_CHAIN_STAMPS
// Diagnostic: wall-clock ticks (100 MHz) at which every workgroup of filter 0 published its head in each of the first 2048
// exchanges of the handle, row 64 = the moment workgroup 0's poll saw all of them.  out: [65][2048].
extern "C" int ekf_debug_exchange_trace(ekf_handle h, long long *out) {
    if (!h || !out) return EKF_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    HIP_TRY(stream_wait(h->s_chain));
    HIP_TRY(hipMemcpy(out, h->dv.dbg + 32, (size_t)65 * 2048 * sizeof(long long), hipMemcpyDeviceToHost));
    return EKF_OK;
}
#endif

// ---- timing ---------------------------------------------------------------------------------------
extern "C" int ekf_timer_start(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    HIP_TRY(hipEventRecord(h->t0, h->s_chain));
    return EKF_OK;
}

extern "C" int ekf_timer_stop(ekf_handle h, double *ms_out) {
    if (!h || !ms_out) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    if (h->overlap && h->prev_pending > 0) HIP_TRY(hipStreamWaitEvent(h->s_chain, h->pass_done[h->ev_idx], 0));  // the pass in flight counts
    HIP_TRY(hipEventRecord(h->t1, h->s_chain));
    HIP_TRY(event_wait(h->t1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->t0, h->t1));
    *ms_out = ms;
    return EKF_OK;
}

extern "C" int ekf_flush_profile(ekf_handle h, int enable) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    h->prof_flush = enable != 0;
    if (h->prof_flush) {
        HIP_TRY(hipSetDevice(h->device));
        return prof_reserve(h, prof_pairs_for_script(h));  // the pool is sized here and in ekf_script_load, not while launches are enqueued
    }
    return EKF_OK;
}

extern "C" int ekf_fused_pass(ekf_handle h) { return h && h->solo_fuse ? 1 : 0; }

extern "C" int ekf_flush_profile_read(ekf_handle h, long long *launches_out, double *total_ms_out) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    { int rc_ = stream_stop(h); if (rc_) return rc_; }  // (a resident streaming launch leaves first)
    HIP_TRY(stream_wait(h->s_chain));
    if (h->overlap) HIP_TRY(stream_wait(h->s_flush));
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->prof_pool[i], h->prof_pool[i + 1]));
        h->prof_ms += ms;
        h->prof_launches++;
    }
    if (h->prof_fused_passes > 0) {
        // k_solo launches were timed whole: report dense passes, not launches -- the time per pass then includes the measurement
        // loop of its window (the pass is not a kernel of its own there; ekf_fused_pass() tells the caller)
        h->prof_launches = h->prof_launches - h->prof_solo_pairs + h->prof_fused_passes;
        h->prof_fused_passes = h->prof_solo_pairs = 0;
    }
    h->prof_used = 0;
    if (launches_out) *launches_out = h->prof_launches;
    if (total_ms_out) *total_ms_out = h->prof_ms;
    h->prof_launches = 0;
    h->prof_ms = 0;
    return EKF_OK;
}
