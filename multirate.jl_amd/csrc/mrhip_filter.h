// mrhip_filter.h -- the filter object behind the opaque mrhip_filter handle (api.hip, arb_schedule.hip, stream_state.hip,
// cascade.cpp).  Kept out of mrhip_internal.h so that a change here does not rebuild the kernel translation units.
#pragma once

#include "mrhip_internal.h"

// ---------------------------------------------------------------------------------------
// the filter object
// ---------------------------------------------------------------------------------------
namespace mrhip {
// zero elements either side of a device tap vector (api.hip upload_taps; kernels_fir_stream.hip plans against it)
constexpr int kTapPad = 256;
}

struct mrhip_filter {
    int kind = 0, th = 0, tx = 0, ty = 0;
    int nc = 1;
    bool r_f64 = false;
    int64_t nch = 1;
    int64_t hLen = 0, L = 1, M = 1, Nphi = 1, T = 1, H = 0;
    int device = 0;
    int num_cus = 256;
    int numerics = MRHIP_NUMERICS_STRICT;
    int force_generic = 0;   // MRHIP_FORCE_GENERIC=1 in the environment: always use the universal kernels

    // device memory
    void *d_taps = nullptr, *d_dtaps = nullptr;            // (inside d_taps_alloc / d_dtaps_alloc: kTapPad zero elements either side)
    void *d_taps_alloc = nullptr, *d_dtaps_alloc = nullptr;
    double *d_pnfb = nullptr;              // FIRFarrow: polynomial filter bank on the device
    double *d_pnfb_t = nullptr;            // ... degree-major and padded to 32 taps, [polyorder+1][32] (farrow_wave_kernel; tapsPerPhi <= 32)
    std::vector<double> h_pnfb;            // ... and on the host, [T][polyorder+1]
    int64_t polyorder = 0;
    void *d_hist[3] = {nullptr, nullptr, nullptr};   // [0], [1]: ping-pong; [2]: zeros(historyLen), read-only -- what a reset() makes current (hist_other)
    unsigned *d_counters = nullptr;   // pair kernel's dynamic scheduling: 33 counters, 256 bytes apart, zero between launches
    int hist_cur = 0;
    // host copies of the taps in tap dtype (for get_taps)
    std::vector<unsigned char> h_taps, h_dtaps;

    // streaming state (1-based, reference field names in multirate_hip.h)
    int64_t phiIdx = 1, inputDeficit = 1, xIdx = 1;
    double rate = 0.0, phiAcc = 1.0, alpha = 0.0, delta = 0.0;

    // the stream state on the device (DevStream above; stream_state.hip).  The host fields above mirror it exactly while
    // mirror_valid; device-planned calls whose result the host has not yet collected (mrhip_filt_device_async, calls
    // captured into a HIP graph) clear the flag, mrhip_sync_state / any call that needs the state restores it.
    mrhip::DevStream *d_rec = nullptr;      // device
    mrhip::DevCall *d_call = nullptr;       // device: the call record of the filter's current device-planned call
    mrhip::DevCall *d_calls[2] = {nullptr, nullptr};   // ... FIRArbitrary / FIRFarrow alternate between two (see s_sched)
    bool async_pending = false;             // asynchronous calls are outstanding (nobody collected their result yet)
    bool last_call_dev_planned = false;   // the latest filt! was planned on the device: its DevCall holds that call's count (what a chained call reads)
    const mrhip::DevCall *last_call_rec = nullptr;   // ... and this is that DevCall (FIRArbitrary / FIRFarrow alternate between two)
    mrhip::DevStream *h_rec = nullptr;      // pinned host memory the plan kernels mirror the record into
    bool mirror_valid = true;
    bool captured = false;                  // a call went into a HIP graph: replays advance the record behind the host's back
    hipEvent_t ev_rec = nullptr;            // recorded behind a plan kernel: the mirror holds that call's result once it fires

    // FIRArbitrary schedule staging
    std::vector<int32_t> sched_n;
    std::vector<double> sched_acc;
    // schedule cache: valid when computed from exactly this (state, x_len); lets next_output_count and
    // the filt call that follows share one evaluation of the serial recurrence
    bool sched_cached = false;
    int64_t sched_xlen = -1, sched_count = 0;
    double sched_acc0 = 0.0;
    int64_t sched_deficit0 = 0;
    mrhip::ArbState sched_end;
    void *pin_n = nullptr, *pin_acc = nullptr;   // pinned host
    size_t pin_cap = 0;
    void *d_sched_n = nullptr, *d_sched_acc = nullptr;
    size_t d_sched_cap = 0;
    hipEvent_t sched_copied = nullptr;
    bool sched_in_flight = false;

    // The schedule of call i is evaluated on a stream of its own, s_sched, so that it runs BESIDE the filter kernel of call
    // i-1 (the record moves on with the schedule's FINISH kernel, long before that call's filter kernel ends): schedule
    // buffers and call records alternate (flip), ev_fin[b] orders the filter kernel behind its schedule, ev_filt[b] the
    // next writer of buffer b behind the filter kernel that read it.  Every write of the record of such a filter goes
    // through s_sched (in program order).  Inside a HIP-graph capture everything stays on the capturing stream.
    hipStream_t s_sched = nullptr;
    // small calls run their schedule on the CALLER's stream instead (api.hip: inline_sched): whatever was enqueued on s_sched since the
    // caller's stream last waited for it (a reset's or set_state's record, a long call's schedule) must be waited for first
    bool sched_dirty = false;
    hipEvent_t ev_sdirty = nullptr;
    hipEvent_t ev_fin[2] = {nullptr, nullptr}, ev_filt[2] = {nullptr, nullptr};
    bool ev_filt_valid[2] = {false, false};
    int flip = 0;
    hipEvent_t ev_chain = nullptr;          // recorded behind a schedule that ran on the CALLER's stream (a chained call): the schedule stream's next use waits for it
    bool chain_pending = false;
    hipStream_t chain_stream = nullptr;     // ... that stream: the event is recorded only when the schedule stream is next used (a stream of small calls never does)
    // memo: the schedule is a pure function of (accumulator, inputDeficit, x_len): a call that repeats the call whose
    // entries buffer memo_buf still holds reuses them (reset + the same block again: benchmarks, batches of equal files)
    bool memo_valid = false;
    int memo_buf = 0;
    double memo_acc0 = 0.0;
    int64_t memo_d0 = 0, memo_xlen = -1, memo_count = 0, memo_per_pos_end = 0;
    mrhip::ArbState memo_end;
    double memo_drift = 0.0, memo_ksteps = 0.0;
    int64_t stat_memo_hits = 0;

    // device-evaluated schedule (arb_schedule.hip, kernels_schedule.hip)
    mrhip::SchedPlan splan{};
    int64_t sched_prefix = 65536, sched_pmax = 1 << 22, sched_device_min = 1 << 16;   // MRHIP_SCHED_* (read at create)
    int sched_corrupt_piece = -1;                 // test hook: falsify the tables of this device piece (counted per filter)
    bool sched_use_cycle = true;
    double sched_drift = 0.0, sched_ksteps = 0.0; // running drift estimate of the stream: (true - un-rounded phase) over ksteps steps
    void *ds_n[2] = {nullptr, nullptr}, *ds_acc[2] = {nullptr, nullptr};   // schedule buffers ([0] in use: one stream, in order)
    size_t ds_cap[2] = {0, 0};
    double *ds_pathT = nullptr;
    int *ds_pathW = nullptr;
    mrhip::SchedGroupEntry *ds_gtab = nullptr;
    mrhip::SchedGroupStart *ds_gstart = nullptr;
    int64_t ds_work_groups = 0;
    mrhip::SchedPieceState *ds_state = nullptr, *ds_pin_state = nullptr;
    int64_t ds_state_cap = 0;
    mrhip::SchedStatus *ds_status = nullptr, *ds_pin_status = nullptr;
    // a detected cycle of the accumulator (PERIODIC mode)
    bool per_valid = false;
    int64_t per_Q = 0, per_XQ = 0, per_pos = 0, per_reset_pos = -1;
    std::vector<double> per_acc;
    std::vector<int64_t> per_xoff;                // [Q + 1] xIdx advance from cycle position 0
    int per_span[mrhip::kSchedSpanSizes] = {};
    double *d_per_acc = nullptr;
    long long *d_per_xoff = nullptr;
    int64_t stat_host_steps = 0, stat_periodic_steps = 0, stat_device_pieces = 0, stat_fallback_pieces = 0;

    // host-pointer path staging: two slots each way (H2D of piece i+1 and D2H of piece i-1 overlap the kernel of
    // piece i: copy streams s_in / s_out beside the kernel stream own_stream, ordered by the events below)
    void *d_xbuf[2] = {nullptr, nullptr}, *d_ybuf[2] = {nullptr, nullptr};
    size_t d_xcap[2] = {0, 0}, d_ycap[2] = {0, 0};
    hipStream_t own_stream = nullptr, s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};

    // Stream ordering of the per-filter device state (history ping-pong, counters, schedule buffers): every call
    // enqueues on the caller's stream; when a call arrives on a DIFFERENT stream than the previous one, an event is
    // recorded on the previous stream and the new stream waits for it (adopt_stream in api.hip).  Nothing is paid while
    // a filter stays on one stream.
    hipStream_t last_stream = nullptr;
    bool last_stream_valid = false;
    hipEvent_t xs_event = nullptr;

    // mrhip_filt_device_multi: descriptor staging of the launches this filter leads (pinned host + device, one event)
    void *multi_pin = nullptr, *multi_dev = nullptr;
    size_t multi_cap = 0;
    hipEvent_t multi_ev = nullptr;
    bool multi_in_flight = false;

    int mod_form = 0;                  // FIRArbitrary / FIRFarrow: 1 = update()'s mod() as Julia Base before 0.4 computed it (mrhip_set_mod_form)
    bool ring_open = false;            // the filter feeds a ring of arriving chunks (ring_api.inc): its own entry points refuse calls meanwhile
    struct mrhip_ring *ring = nullptr; // ... that ring (mrhip_destroy shuts it down first)
    // what a closed ring leaves behind for the next one of this filter (pinned descriptor ring, device-side ring, history slots, stream):
    // allocating and freeing them was 0.45 of the 0.6 ms it cost to open and close a ring (hipFree waits for the device)
    struct RingCache { void *host = nullptr, *host_dev = nullptr, *dev = nullptr, *hist = nullptr; size_t slot_bytes = 0; hipStream_t stream = nullptr; } ring_cache;

    // measurement
    bool timing = false;
    int timing_stride = 1;            // bracket every timing_stride-th compute launch (1 = all)
    int timing_group = 1;             // > 1: one bracket around every timing_group consecutive compute launches
    int64_t timing_launch = 0;        // compute launches seen since timing was enabled
    bool timing_open = false;         // between the two marks of a launch
    bool ev_skip = false;             // ... of a launch that is not bracketed
    std::vector<hipEvent_t> ev_pool;   // pairs: [2i] start, [2i+1] stop
    size_t ev_used = 0;                // events handed out since the last mrhip_timing_read
    const char *last_kernel = "";
};

// ---------------------------------------------------------------------------------------
// what works on the filter object: the schedule of a call (arb_schedule.hip), the device record (stream_state.hip)
// ---------------------------------------------------------------------------------------
namespace mrhip {
void sched_configure(mrhip_filter *f);
void sched_forget(mrhip_filter *f);
void sched_free(mrhip_filter *f);
bool sched_wants_device(const mrhip_filter *f, int64_t est);
// Enqueue the schedule of one call on `s` (see arb_schedule.hip).  host_ok: the host's copy of the stream state is exact
// and may be used (serial prefix, closed form of a cycle); otherwise everything is taken from the device record.
struct SchedOut {
    int buf = 0;                  // schedule buffer that will hold the entries
    bool host_known = false;      // the host evaluated the whole call itself: count / end are final, nothing to collect
    bool pending = false;         // a FINISH kernel will deliver the result into the mirror (ev_rec)
    int64_t count = 0;
    ArbState end;
    bool periodic = false;
    int64_t per_pos_end = 0;
    double drift = 0.0, ksteps = 0.0;
    // for a host redo after a failed piece: what was enqueued
    std::vector<int64_t> pk0, psteps;
    int64_t k_first = 0;
    bool memo_hit = false;        // the entries of an identical earlier call were reused (arb_schedule.hip: memo)
    double memo_acc0 = 0.0;       // the call-start state, for the memo entry this call becomes
    int64_t memo_d0 = 0;
};
int sched_enqueue(mrhip_filter *f, int64_t x_len, int64_t est, int64_t y_capacity, long long *count_out, bool host_ok, hipStream_t s, SchedOut *out,
                  const mrhip::DevCall *x_from = nullptr);
int sched_collect(mrhip_filter *f, int64_t x_len, int64_t est, int64_t y_capacity, long long *count_out, hipStream_t s, SchedOut *io, bool *relaunch);
// stream_state.hip: the device record
int rec_alloc(mrhip_filter *f);
void rec_free(mrhip_filter *f);
// write the host's state into the device record (and its mirror), in stream order on `s`; call_n_out >= 0 also arms the
// DevCall with that output count (a call the host evaluated, whose filter kernel reads the DevCall)
int rec_push(mrhip_filter *f, hipStream_t s, long long call_n_out = -1, long long n_written = -1, unsigned *zero_counters = nullptr);
// wait for everything enqueued on the filter's behalf and take the device record over into the host fields
int rec_pull(mrhip_filter *f);
// api.hip: is a stream the library has been called on still being captured into a graph?
bool any_capture_active();
hipError_t launch_poly_plan(mrhip_filter *f, int64_t x_len, long long P, long long y_capacity, long long *count_out, hipStream_t s, const DevCall *x_from = nullptr);
}  // namespace mrhip
// hipDeviceSynchronize as the library uses it: the wait for work on streams it never saw (replays of a graph that holds a filter's
// calls; whatever ran on a stream that no longer exists).  While ANOTHER stream of the process is being captured in the global mode
// (torch.cuda.graph's default) a plain hipDeviceSynchronize from any thread is an "unsafe call" and invalidates that capture: the
// calling thread switches to the relaxed mode for the one call (mrhip_next_output_count on a filter that was once captured,
// issued while another filter's stream was being captured, killed the capture).
inline hipError_t device_sync_relaxed()
{
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    const hipError_t e = hipDeviceSynchronize();
    if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
    return e;
}

// The history slot the next call WRITES (the slots 0 and 1 ping-pong; slot 2 holds zeros and is what reset() makes current instead of
// zeroing a slot: read-only until a captured call adopts it as its in-place slot -- after which reset() zeroes the current slot itself)
inline int hist_other(const mrhip_filter *f) { return f->hist_cur == 2 ? 0 : f->hist_cur ^ 1; }
// ... and before the CURRENT slot is written in place (set_history, a ring handing the stream back): never the zeros
inline void hist_leave_zeros(mrhip_filter *f) { if (f->hist_cur == 2) f->hist_cur = 0; }

// ring_api.inc: everything mrhip_ring_close does except freeing the handle (mrhip_destroy: a filter that still feeds a ring)
namespace mrhip { int ring_shutdown(struct mrhip_ring *r); void ring_cache_free(mrhip_filter *f); }

