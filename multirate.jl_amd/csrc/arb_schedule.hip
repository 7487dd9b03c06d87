// arb_schedule.hip -- host side of the device-evaluated phase schedule of FIRArbitrary / FIRFarrow
// (update(), src/Filters.jl:663-673 and :780-792; device side: kernels_schedule.hip; model: scripts/sched_model.py).
//
// One filt! call of x_len samples needs, per output k, the input index n_k and the accumulator acc_k.  Three ways:
//   HOST      the serial loop of host_logic.cpp (1.3 ns per output): short calls, and whatever the device path does
//             not cover (rates above ~10 N/32 need more candidates per segment than the tables kernel holds).
//   PERIODIC  the recurrence is a deterministic map of acc alone, so when the state after the serial prefix equals
//             -- bit for bit -- a state inside the prefix, the schedule repeats from there on for ever: closed form,
//             emitted by one trivial kernel, counted on the host, no synchronisation.  (Rates such as 1.0, 3.0, 11/7:
//             exactly the ones whose phase sits ON the wrap / binade thresholds, where nothing but exact states helps.)
//   TABLES    everything else: kernels_schedule.hip, piece by piece (sizes grow with the length of the drift
//             baseline), one host synchronisation (of the caller's stream) per call for the count; a piece whose
//             verification fails is redone by the host loop.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mrhip_filter.h"

namespace mrhip {
namespace {

constexpr int kPerThreads = 256;

// PERIODIC: entry j of the call (j >= 0) is cycle position (pos + j) mod Q.  call != NULL: pos, xbase and n come from
// the DevCall the plan kernel below filled (a device-planned call; the arguments then hold upper bounds).
__global__ __launch_bounds__(kPerThreads) void sched_periodic_kernel(const double *__restrict__ per_acc, const long long *__restrict__ per_xoff,
                                                                     long long Q, long long XQ, long long pos, long long xbase,
                                                                     long long n, int *__restrict__ sched_n, double *__restrict__ sched_acc,
                                                                     const DevCall *__restrict__ call)
{
    if (call) { pos = call->per_pos; xbase = call->per_xbase; n = call->k_done; }
    for (long long j = blockIdx.x * static_cast<long long>(kPerThreads) + threadIdx.x; j < n; j += static_cast<long long>(gridDim.x) * kPerThreads) {
        const long long t = pos + j, cyc = t / Q, r = t - cyc * Q;
        sched_n[j] = static_cast<int>(xbase + cyc * XQ + per_xoff[r]);
        sched_acc[j] = per_acc[r];
    }
}

// The closed form of a cycle evaluated where the state lives: output count by bisection on x_j (never decreasing), end
// state, record advanced.  A stream that is not where the cycle says it is (somebody moved it) cannot be served this
// way and has nobody to fall back to in a call that is not waited for: the error is left in the record.
struct PerPlanArgs {
    DevStream *rec, *mirror;
    DevCall *call;
    const double *per_acc;
    const long long *per_xoff;
    long long *count_out;
    long long Q, XQ, x_len, est, y_capacity;
    const DevCall *x_from;        // a chained call: the input length is this record's count (x_len: its upper bound)
};
__global__ __launch_bounds__(64) void sched_periodic_plan_kernel(PerPlanArgs a)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (a.x_from) {
        const long long n = a.x_from->n_out;
        a.x_len = n < 0 ? 0 : (n < a.x_len ? n : a.x_len);
    }
    DevStream r = *a.rec;
    DevCall c{};
    c.x_len = a.x_len;
    r.sched_fail = kSchedNoFail;
    const long long pos = r.per_pos;
    if (!(pos >= 0 && pos < a.Q) || !(a.per_acc[pos] == r.acc)) {
        r.error = MRHIP_ERR_INVALID_ARG;
        r.n_written = 0;
        r.calls += 1;
    } else {
        const long long x0 = r.inputDeficit, xb = x0 - a.per_xoff[pos];
        auto per_x = [&](long long j) { const long long t = pos + j, cyc = t / a.Q, q = t - cyc * a.Q; return xb + cyc * a.XQ + a.per_xoff[q]; };
        long long lo = 0, hi = (a.est > 0 ? a.est : 0) + 2;
        while (per_x(hi) <= a.x_len) hi *= 2;
        while (lo < hi) {
            const long long mid = lo + (hi - lo) / 2;
            if (per_x(mid) > a.x_len) hi = mid; else lo = mid + 1;
        }
        const long long m = lo;
        c.n_out = m;
        if (m > a.y_capacity) { c.n_out = a.y_capacity; r.error = MRHIP_ERR_BUFFER_TOO_SMALL; }
        if (m > a.est) { c.n_out = 0; r.error = MRHIP_ERR_INVALID_ARG; }      // est bounds the count: cannot happen
        c.k_done = c.n_out;
        c.per_pos = pos; c.per_xbase = xb;
        r.inputDeficit = per_x(m) - a.x_len;                                  // Filters.jl:734
        r.per_pos = (pos + m) % a.Q;
        r.acc = a.per_acc[r.per_pos];
        r.n_written = c.n_out;
        r.calls += 1;
    }
    *a.call = c;
    *a.rec = r;
    *a.mirror = r;
    if (a.count_out) *a.count_out = c.n_out;
}

int64_t env_i64(const char *name, int64_t dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoll(v) : dflt;
}

double wrap_half(double d, double N)
{
    if (d > 0.5 * N) return d - N;
    if (d < -0.5 * N) return d + N;
    return d;
}

// Allocations wait for the stream and are not capturable: inside a HIP-graph capture a buffer that would have to grow is
// refused BEFORE anything touches the runtime (an allocation call there would invalidate the capture) -- the caller makes one
// plain call of the size first, as with every graph-captured library.
bool capturing(hipStream_t st)
{
    if (!st) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs == hipStreamCaptureStatusActive;
}
int refuse_growth() { return fail(MRHIP_ERR_UNSUPPORTED, "the schedule's work buffers are allocated by the first call of a size and an allocation cannot be captured: make one plain call of this size before the capture"); }

int ensure_pinned(mrhip_filter *f, size_t n, hipStream_t st)
{
    if (n <= f->pin_cap) return MRHIP_OK;
    if (capturing(st)) return refuse_growth();
    const size_t cap = std::max<size_t>(n + n / 4, 4096);
    if (f->sched_in_flight) { MRHIP_CHECK_HIP(hipEventSynchronize(f->sched_copied)); f->sched_in_flight = false; }
    MRHIP_CHECK_HIP(hipStreamSynchronize(st));
    if (f->pin_n) (void)hipHostFree(f->pin_n);
    if (f->pin_acc) (void)hipHostFree(f->pin_acc);
    f->pin_n = f->pin_acc = nullptr;
    f->pin_cap = 0;
    MRHIP_CHECK_HIP(hipHostMalloc(&f->pin_n, cap * sizeof(int32_t), hipHostMallocDefault));
    MRHIP_CHECK_HIP(hipHostMalloc(&f->pin_acc, cap * sizeof(double), hipHostMallocDefault));
    f->pin_cap = cap;
    return MRHIP_OK;
}

int ensure_ds(mrhip_filter *f, int b, size_t n, hipStream_t st)
{
    if (n <= f->ds_cap[b]) return MRHIP_OK;
    if (capturing(st)) return refuse_growth();
    const size_t cap = n + n / 8 + 4096;
    MRHIP_CHECK_HIP(hipStreamSynchronize(st));
    if (f->ds_n[b]) (void)hipFree(f->ds_n[b]);
    if (f->ds_acc[b]) (void)hipFree(f->ds_acc[b]);
    f->ds_n[b] = f->ds_acc[b] = nullptr;
    f->ds_cap[b] = 0;
    MRHIP_CHECK_HIP(hipMalloc(&f->ds_n[b], cap * sizeof(int32_t)));
    MRHIP_CHECK_HIP(hipMalloc(&f->ds_acc[b], cap * sizeof(double)));
    f->ds_cap[b] = cap;
    return MRHIP_OK;
}

int ensure_work(mrhip_filter *f, int64_t groups, int64_t pieces, hipStream_t st)
{
    const SchedPlan &c = f->splan;
    if ((groups > f->ds_work_groups || pieces + 2 > f->ds_state_cap || !f->ds_status) && capturing(st)) return refuse_growth();
    if (groups > f->ds_work_groups) {
        MRHIP_CHECK_HIP(hipStreamSynchronize(st));
        for (void *p : {static_cast<void *>(f->ds_pathT), static_cast<void *>(f->ds_pathW), static_cast<void *>(f->ds_gtab), static_cast<void *>(f->ds_gstart)})
            if (p) (void)hipFree(p);
        f->ds_pathT = nullptr; f->ds_pathW = nullptr; f->ds_gtab = nullptr; f->ds_gstart = nullptr;
        f->ds_work_groups = 0;
        const size_t segs = static_cast<size_t>(groups) * 64;
        MRHIP_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&f->ds_pathT), segs * c.nwin * sizeof(double)));
        MRHIP_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&f->ds_pathW), segs * c.nwin * sizeof(int)));
        MRHIP_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&f->ds_gtab), static_cast<size_t>(groups) * c.nwin * sizeof(SchedGroupEntry)));
        MRHIP_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&f->ds_gstart), static_cast<size_t>(groups + 1) * sizeof(SchedGroupStart)));
        f->ds_work_groups = groups;
    }
    if (pieces + 2 > f->ds_state_cap) {
        MRHIP_CHECK_HIP(hipStreamSynchronize(st));
        if (f->ds_state) (void)hipFree(f->ds_state);
        if (f->ds_pin_state) (void)hipHostFree(f->ds_pin_state);
        f->ds_state = nullptr; f->ds_pin_state = nullptr; f->ds_state_cap = 0;
        const int64_t cap = pieces + 64;
        MRHIP_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&f->ds_state), static_cast<size_t>(cap) * sizeof(SchedPieceState)));
        MRHIP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&f->ds_pin_state), static_cast<size_t>(cap) * sizeof(SchedPieceState), hipHostMallocDefault));
        f->ds_state_cap = cap;
    }
    if (!f->ds_status) {
        MRHIP_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&f->ds_status), sizeof(SchedStatus)));
        MRHIP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&f->ds_pin_status), 2 * sizeof(SchedStatus), hipHostMallocDefault));
    }
    return MRHIP_OK;
}

// upload host entries [0, cnt) of the pinned staging to outputs [k0, k0 + cnt) of schedule buffer b
int upload_entries(mrhip_filter *f, int b, int64_t k0, int64_t cnt, hipStream_t st)
{
    if (cnt <= 0) return MRHIP_OK;
    MRHIP_CHECK_HIP(hipMemcpyAsync(static_cast<int32_t *>(f->ds_n[b]) + k0, f->pin_n, static_cast<size_t>(cnt) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    MRHIP_CHECK_HIP(hipMemcpyAsync(static_cast<double *>(f->ds_acc[b]) + k0, f->pin_acc, static_cast<size_t>(cnt) * sizeof(double), hipMemcpyHostToDevice, st));
    MRHIP_CHECK_HIP(hipEventRecord(f->sched_copied, st));
    f->sched_in_flight = true;
    return MRHIP_OK;
}

int wait_pinned_free(mrhip_filter *f)
{
    if (f->sched_in_flight) { MRHIP_CHECK_HIP(hipEventSynchronize(f->sched_copied)); f->sched_in_flight = false; }
    return MRHIP_OK;
}

// x of entry j of a periodic call that starts at cycle position pos with xIdx = x0
inline int64_t per_x(const mrhip_filter *f, int64_t x0, int64_t j)
{
    const int64_t t = f->per_pos + j, cyc = t / f->per_Q, r = t - cyc * f->per_Q;
    return x0 - f->per_xoff[static_cast<size_t>(f->per_pos)] + cyc * f->per_XQ + f->per_xoff[static_cast<size_t>(r)];
}

// the state after the prefix equals a state inside it?  (host entries [0, cnt) in the pinned staging; `st` after them)
int try_find_cycle(mrhip_filter *f, int64_t cnt, const ArbState &st, hipStream_t stream)
{
    const double *acc = static_cast<const double *>(f->pin_acc);
    const int32_t *n = static_cast<const int32_t *>(f->pin_n);
    int64_t j = cnt - 1;
    const int64_t lo = std::max<int64_t>(0, cnt - (1 << 16));
    while (j >= lo && acc[j] != st.acc) --j;
    if (j < lo) return MRHIP_OK;
    const int64_t Q = cnt - j;
    f->per_acc.assign(acc + j, acc + cnt);
    f->per_xoff.assign(static_cast<size_t>(Q) + 1, 0);
    for (int64_t i = 0; i < Q; ++i) {
        const int64_t nxt = i + 1 < Q ? n[j + i + 1] : st.xIdx;
        f->per_xoff[static_cast<size_t>(i) + 1] = f->per_xoff[static_cast<size_t>(i)] + (nxt - n[j + i]);
    }
    f->per_XQ = f->per_xoff[static_cast<size_t>(Q)];
    if (f->per_XQ < 1) return MRHIP_OK;                 // (cannot happen: a cycle of the phase wraps at least once)
    f->per_Q = Q;
    f->per_pos = 0;
    f->per_reset_pos = -1;                              // where reset()'s state (acc == 1.0) sits on the cycle, if it does
    for (int64_t i = 0; i < Q; ++i)
        if (f->per_acc[static_cast<size_t>(i)] == 1.0) { f->per_reset_pos = i; break; }
    for (int z = 0; z < kSchedSpanSizes; ++z) {          // largest advance over (64 << z) - 1 steps, from any cycle position
        const int64_t m = (static_cast<int64_t>(kSchedSpanBase) << z) - 1, full = m / Q, rem = m - full * Q;
        int64_t best = 0;
        for (int64_t i = 0; i < Q; ++i) {
            const int64_t e = i + rem;
            const int64_t adv = e <= Q ? f->per_xoff[static_cast<size_t>(e)] - f->per_xoff[static_cast<size_t>(i)]
                                       : f->per_XQ - f->per_xoff[static_cast<size_t>(i)] + f->per_xoff[static_cast<size_t>(e - Q)];
            best = std::max(best, adv);
        }
        f->per_span[z] = static_cast<int>(std::min<int64_t>(full * f->per_XQ + best, 0x7fffffff));
    }
    MRHIP_CHECK_HIP(hipStreamSynchronize(stream));
    if (f->d_per_acc) (void)hipFree(f->d_per_acc);
    if (f->d_per_xoff) (void)hipFree(f->d_per_xoff);
    f->d_per_acc = nullptr; f->d_per_xoff = nullptr;
    MRHIP_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&f->d_per_acc), static_cast<size_t>(Q) * sizeof(double)));
    MRHIP_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&f->d_per_xoff), static_cast<size_t>(Q + 1) * sizeof(long long)));
    static_assert(sizeof(long long) == sizeof(int64_t), "per_xoff is uploaded as long long");
    MRHIP_CHECK_HIP(hipMemcpy(f->d_per_acc, f->per_acc.data(), static_cast<size_t>(Q) * sizeof(double), hipMemcpyHostToDevice));
    MRHIP_CHECK_HIP(hipMemcpy(f->d_per_xoff, f->per_xoff.data(), static_cast<size_t>(Q + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    f->per_valid = true;
    return MRHIP_OK;
}

}  // namespace

void sched_configure(mrhip_filter *f)
{
    f->splan = make_sched_plan(f->delta, f->Nphi, static_cast<int>(env_i64("MRHIP_SCHED_WIN_MULT", 1)), static_cast<int>(env_i64("MRHIP_SCHED_WIN_MIN", 4)));
    f->sched_prefix = std::max<int64_t>(env_i64("MRHIP_SCHED_PREFIX", 65536), 0) / kSchedGroup * kSchedGroup;
    f->sched_pmax = std::max<int64_t>(env_i64("MRHIP_SCHED_PMAX", 1 << 23) / kSchedGroup * kSchedGroup, kSchedGroup);
    // below it a call that is waited for takes the host loop: 60 us + 1.3 ns per output against 75 us for the kernels of one piece
    // and the wait (profiles/r04/experiments.md Q; 2^16 until the baseline of short calls was kept)
    f->sched_device_min = env_i64("MRHIP_SCHED_DEVICE_MIN", 1 << 14);
    f->sched_corrupt_piece = static_cast<int>(env_i64("MRHIP_SCHED_CORRUPT", -1));
    if (env_i64("MRHIP_SCHED_DEVICE", 1) == 0) f->splan.ok = 0;
    f->sched_use_cycle = env_i64("MRHIP_SCHED_CYCLE", 1) != 0;
}

// the stream position was set from outside (reset, set_state): the drift estimate and a detected cycle no longer apply
void sched_forget(mrhip_filter *f)
{
    f->sched_drift = 0.0;
    f->sched_ksteps = 0.0;
    f->per_valid = false;
}

void sched_free(mrhip_filter *f)
{
    for (int b = 0; b < 2; ++b) {
        if (f->ds_n[b]) (void)hipFree(f->ds_n[b]);
        if (f->ds_acc[b]) (void)hipFree(f->ds_acc[b]);
    }
    for (void *p : {static_cast<void *>(f->ds_pathT), static_cast<void *>(f->ds_pathW), static_cast<void *>(f->ds_gtab), static_cast<void *>(f->ds_gstart),
                    static_cast<void *>(f->ds_state), static_cast<void *>(f->ds_status), static_cast<void *>(f->d_per_acc), static_cast<void *>(f->d_per_xoff)})
        if (p) (void)hipFree(p);
    if (f->ds_pin_state) (void)hipHostFree(f->ds_pin_state);
    if (f->ds_pin_status) (void)hipHostFree(f->ds_pin_status);
}

bool sched_wants_device(const mrhip_filter *f, int64_t est)
{
    // (the device evaluation implements the exact remainder: the older mod() form differs from it unless N𝜙 is a power of two)
    if (f->mod_form != 0 && (f->Nphi & (f->Nphi - 1)) != 0) return false;
    return f->splan.ok && est >= f->sched_device_min && est < 0x7fffffffLL;
}

static DevStream *mirror_of(mrhip_filter *f)
{
    void *p = nullptr;
    if (hipHostGetDevicePointer(&p, f->h_rec, 0) != hipSuccess) { (void)hipGetLastError(); return f->h_rec; }
    return static_cast<DevStream *>(p);
}
static SchedPieceState *fail_state_of(mrhip_filter *f)      // pinned, behind the mirror (rec_alloc: 512 bytes)
{
    return reinterpret_cast<SchedPieceState *>(reinterpret_cast<unsigned char *>(mirror_of(f)) + 256);
}
static const SchedPieceState *fail_state_host(const mrhip_filter *f)
{
    return reinterpret_cast<const SchedPieceState *>(reinterpret_cast<const unsigned char *>(f->h_rec) + 256);
}

// TABLES from schedule entry k with the piece list sized by the drift baseline `ks`: begin, pieces, finish, event.
// host state != NULL: the call-start state of the first piece is the host's (behind a prefix or a redone piece).
static int enqueue_tables(mrhip_filter *f, int64_t x_len, int64_t est, int64_t y_capacity, long long *count_out, int64_t k, double ks,
                          const SchedPieceState *host_state, bool serial_fallback, hipStream_t s, SchedOut *out, const DevCall *x_from = nullptr)
{
    const SchedPlan &c = f->splan;
    const int b = out->buf;
    out->pk0.clear(); out->psteps.clear();
    out->k_first = k;
    int64_t kk = k;
    double kz = ks;
    if (c.ok)
        while (kk <= est) {
            // a piece may be 16x as long as the drift baseline behind it: the slope error then moves the last segments'
            // starts out of the candidate window, where the shift argument takes over (exact all the same; rates whose
            // phase sits ON a threshold -- where it would not hold -- cycle, and take the closed form)
            int64_t P = sched_piece_steps(kz, f->sched_pmax);
            // the piece that reaches the bound of the count is cut to the groups it needs (always the last one; the FINISH
            // kernel, which re-derives the sizes, only ever sums the pieces in front of it)
            if (kk + P > est + 1) P = std::max<int64_t>((est + 1 - kk + kSchedGroup - 1) / kSchedGroup, 1) * kSchedGroup;
            out->pk0.push_back(kk); out->psteps.push_back(P);
            kk += P; kz += static_cast<double>(P);
        }
    const int64_t np = static_cast<int64_t>(out->pk0.size());
    if (int rc = ensure_work(f, f->sched_pmax / kSchedGroup, np, s)) return rc;
    if (std::max(kk, est + 1) > static_cast<int64_t>(f->ds_cap[b])) return fail(MRHIP_ERR_INVALID_ARG, "schedule buffer too small (internal)");
    SchedBeginArgs ba{};
    ba.rec = f->d_rec; ba.status = f->ds_status; ba.state = f->ds_state;
    if (host_state) { ba.use_host = 1; ba.acc = host_state->acc; ba.xIdx = host_state->xIdx; ba.drift = host_state->drift; ba.ksteps = host_state->ksteps; }
    ba.x_from = x_from;
    SchedFinishArgs fa{};
    fa.rec = f->d_rec; fa.mirror = mirror_of(f); fa.call = f->d_calls[b];
    fa.status = f->ds_status; fa.state = f->ds_state; fa.fail_state = fail_state_of(f);
    fa.sched_n = static_cast<int *>(f->ds_n[b]); fa.sched_acc = static_cast<double *>(f->ds_acc[b]);
    fa.count_out = count_out;
    fa.k_first = k; fa.ks_first = ks; fa.pmax = f->sched_pmax; fa.est = est + 1; fa.x_len = x_len; fa.y_capacity = y_capacity;
    fa.np = static_cast<int>(np); fa.serial_fallback = serial_fallback ? 1 : 0;
    // BEGIN in the first piece's tables kernel, FINISH in the last piece's emit kernel (MRHIP_SCHED_FUSE=0: launches of their own)
    const bool fuse_on = env_i64("MRHIP_SCHED_FUSE", 1) != 0;
    const bool fuse = fuse_on && np > 0;
    if (!fuse) MRHIP_CHECK_HIP(launch_sched_begin(ba, x_len, k, s));
    for (int64_t p = 0; p < np; ++p) {
        SchedPieceArgs a{};
        a.state = f->ds_state; a.status = f->ds_status;
        a.pathT = f->ds_pathT; a.pathW = f->ds_pathW; a.gtab = f->ds_gtab; a.gstart = f->ds_gstart;
        a.sched_n = static_cast<int *>(f->ds_n[b]); a.sched_acc = static_cast<double *>(f->ds_acc[b]);
        a.k0 = out->pk0[static_cast<size_t>(p)]; a.x_len = x_len;
        a.piece = static_cast<int>(p); a.ngroups = static_cast<int>(out->psteps[static_cast<size_t>(p)] / kSchedGroup);
        a.corrupt_group = f->sched_corrupt_piece == f->stat_device_pieces + p ? 0 : -1;
        if (f->sched_corrupt_piece <= -2) a.corrupt_group = f->sched_corrupt_piece;   // test hook: group -v-2 of every piece starts late
        SchedFuseArgs fu{};
        if (fuse) {
            fu.begin = p == 0; fu.finish = p + 1 == np;
            fu.b = ba; fu.b_x_len = x_len; fu.b_k_first = k;
            fu.f = fa;
        }
        MRHIP_CHECK_HIP(launch_schedule_piece(c, a, s, fuse ? &fu : nullptr));
    }
    if (!fuse) MRHIP_CHECK_HIP(launch_sched_finish(c, fa, s));
    // (the event is what sched_collect waits for: a call nobody collects -- asynchronous, captured: serial_fallback -- does without it, i.e.
    //  without one host call per small call and one node per call of a captured graph)
    if (!serial_fallback) MRHIP_CHECK_HIP(hipEventRecord(f->ev_rec, s));
    out->pending = true;
    return MRHIP_OK;
}

// Enqueue the schedule of one call (x_len samples, at most `est` outputs) into schedule buffer out->buf, everything on the
// CALLER's stream `s`, in front of the filter kernel that reads it: one queue, no events between producer and consumer.
// (A stream of its own for the schedule bought nothing: the previous call's persistent filter kernel holds every CU until it
// ends, profiles/r03/experiments.md B.)  The count and the end state either are final on return (out->host_known: the host
// evaluated the call itself -- a serial prefix that reached the end, the closed form of a cycle) or arrive in the pinned
// mirror behind f->ev_rec (out->pending); either way the device record and the DevCall are current in stream order.
int sched_enqueue(mrhip_filter *f, int64_t x_len, int64_t est, int64_t y_capacity, long long *count_out, bool host_ok, hipStream_t s, SchedOut *out,
                  const DevCall *x_from)
{
    const SchedPlan &c = f->splan;
    const int b = out->buf;              // schedule buffer and call record of this call (the caller alternates them)
    // (allocations wait for the stream: never inside a capture -- the caller warms a filter up before it captures, or the
    //  buffers of an earlier call of the same size are there already)
    if (int rc = ensure_ds(f, b, static_cast<size_t>(est + f->sched_pmax + f->sched_prefix + 2 * kSchedGroup), s)) return rc;
    if (int rc = ensure_pinned(f, static_cast<size_t>(std::max<int64_t>(f->sched_prefix, f->sched_pmax) + kSchedGroup), s)) return rc;
    if (int rc = ensure_work(f, f->sched_pmax / kSchedGroup, 64, s)) return rc;

    if (!host_ok) {
        // ---- everything from the device record ------------------------------------------------------------------
        if (f->memo_valid && f->memo_buf == b) f->memo_valid = false;
        if (f->per_valid) {
            PerPlanArgs pa{};
            pa.rec = f->d_rec; pa.mirror = mirror_of(f); pa.call = f->d_calls[b];
            pa.per_acc = f->d_per_acc; pa.per_xoff = f->d_per_xoff; pa.count_out = count_out;
            pa.Q = f->per_Q; pa.XQ = f->per_XQ; pa.x_len = x_len; pa.est = est; pa.y_capacity = y_capacity;
            pa.x_from = x_from;
            hipLaunchKernelGGL(sched_periodic_plan_kernel, dim3(1), dim3(64), 0, s, pa);
            MRHIP_CHECK_HIP(hipGetLastError());
            if (est + 2 > static_cast<int64_t>(f->ds_cap[b])) return fail(MRHIP_ERR_INVALID_ARG, "schedule buffer too small for the periodic schedule (internal)");
            const long long blocks = std::min<long long>((est + kPerThreads) / kPerThreads, 4096);
            hipLaunchKernelGGL(sched_periodic_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kPerThreads), 0, s, f->d_per_acc, f->d_per_xoff,
                               static_cast<long long>(f->per_Q), static_cast<long long>(f->per_XQ), 0LL, 0LL, 0LL,
                               static_cast<int *>(f->ds_n[b]), static_cast<double *>(f->ds_acc[b]), static_cast<const DevCall *>(f->d_calls[b]));
            MRHIP_CHECK_HIP(hipGetLastError());
            out->pending = true;                 // (nobody collects a call planned from the device record: no event)
            out->periodic = true;
            return MRHIP_OK;
        }
        // the host's lower bound of the drift baseline sizes the pieces (every piece is verified whatever its size)
        return enqueue_tables(f, x_len, est, y_capacity, count_out, 0, f->sched_ksteps, nullptr, true, s, out, x_from);
    }
    if (x_from) return fail(MRHIP_ERR_INVALID_ARG, "a chained call is planned on the device (internal)");

    static const bool prof = env_i64("MRHIP_DEBUG", 0) == 2;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = prof ? now() : 0.0;
    ArbState st{f->phiAcc, f->phiIdx, f->alpha, f->inputDeficit, f->inputDeficit};   // xIdx starts at inputDeficit (Filters.jl:715)
    double drift = f->sched_drift, ksteps = f->sched_ksteps;
    int64_t k = 0;
    bool done = false;
    auto finish_host_state = [&]() {
        st.phiIdx = static_cast<int64_t>(std::floor(st.acc));
        st.alpha = st.acc - static_cast<double>(st.phiIdx);
        st.inputDeficit = st.xIdx - x_len;
    };
    // memo: the schedule is a pure function of (accumulator, inputDeficit, x_len) -- a call that repeats the one whose entries
    // a schedule buffer still holds (reset + the same block again) takes them as they are
    static const bool memo_on = env_i64("MRHIP_SCHED_MEMO", 1) != 0;
    if (memo_on && f->memo_valid && f->memo_acc0 == f->phiAcc && f->memo_d0 == f->inputDeficit && f->memo_xlen == x_len && f->sched_corrupt_piece == -1) {
        out->buf = f->memo_buf;
        out->host_known = true; out->count = f->memo_count; out->end = f->memo_end;
        out->drift = f->memo_drift; out->ksteps = f->memo_ksteps; out->per_pos_end = f->memo_per_pos_end;
        out->periodic = f->per_valid;
        out->memo_hit = true;
        ++f->stat_memo_hits;
        return MRHIP_OK;
    }
    if (f->memo_valid && f->memo_buf == b) f->memo_valid = false;     // buffer b is about to be rewritten
    out->memo_acc0 = f->phiAcc; out->memo_d0 = f->inputDeficit;
    if (x_len < f->inputDeficit) {          // Filters.jl:705-709: not one output
        st.inputDeficit = f->inputDeficit - x_len;
        st.xIdx = f->inputDeficit;
        out->host_known = true; out->count = 0; out->end = st; out->drift = drift; out->ksteps = ksteps;
        out->per_pos_end = f->per_pos;
        return MRHIP_OK;
    }

    // a cycle found earlier still applies only if the stream is exactly where the cycle says it is
    if (f->per_valid && !(f->per_acc[static_cast<size_t>(f->per_pos)] == st.acc)) f->per_valid = false;

    // ---- serial prefix on the host: short baseline for the drift estimate, cycle detection -----------------
    if (!f->per_valid && ksteps < static_cast<double>(f->sched_prefix)) {
        if (int rc = wait_pinned_free(f)) return rc;
        const int64_t want = (f->sched_prefix - static_cast<int64_t>(ksteps) + kSchedGroup - 1) / kSchedGroup * kSchedGroup;
        const double acc_start = st.acc;
        const int64_t cnt = run_arbitrary_schedule_piece(st, f->delta, f->Nphi, x_len, static_cast<int32_t *>(f->pin_n), static_cast<double *>(f->pin_acc), want, &done, f->mod_form);
        if (!done && f->sched_use_cycle)
            if (int rc = try_find_cycle(f, cnt, st, s)) return rc;
        if (int rc = upload_entries(f, b, 0, cnt, s)) return rc;
        // (also when the prefix reached the call's end: `cnt` updates ran either way.  Round 3 kept the baseline only for a
        //  prefix the call outlived, so a stream of calls shorter than the prefix never left the host's loop.)
        drift += wrap_half(st.acc - sched_anchor_host(c, acc_start, static_cast<double>(cnt)), c.N);
        ksteps += static_cast<double>(cnt);
        k = cnt;
        f->stat_host_steps += cnt;
    }

    // ---- PERIODIC: closed form ------------------------------------------------------------------------------
    if (!done && f->per_valid) {
        const int64_t x0 = st.xIdx;
        // first j with x_j > x_len (x_j never decreases): the number of further outputs
        int64_t lo = 0, hi = std::max<int64_t>(est - k, 0) + 2;
        while (per_x(f, x0, hi) <= x_len) hi *= 2;
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (per_x(f, x0, mid) > x_len) hi = mid; else lo = mid + 1;
        }
        const int64_t m = lo;
        if (k + m > static_cast<int64_t>(f->ds_cap[b])) return fail(MRHIP_ERR_INVALID_ARG, "schedule buffer too small for the periodic schedule (internal)");
        if (m > 0) {
            const long long blocks = std::min<long long>((m + kPerThreads - 1) / kPerThreads, 4096);
            hipLaunchKernelGGL(sched_periodic_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kPerThreads), 0, s, f->d_per_acc, f->d_per_xoff,
                               static_cast<long long>(f->per_Q), static_cast<long long>(f->per_XQ), static_cast<long long>(f->per_pos),
                               static_cast<long long>(x0 - f->per_xoff[static_cast<size_t>(f->per_pos)]), static_cast<long long>(m),
                               static_cast<int *>(f->ds_n[b]) + k, static_cast<double *>(f->ds_acc[b]) + k, static_cast<const DevCall *>(nullptr));
            MRHIP_CHECK_HIP(hipGetLastError());
        }
        st.xIdx = per_x(f, x0, m);
        out->per_pos_end = (f->per_pos + m) % f->per_Q;
        st.acc = f->per_acc[static_cast<size_t>(out->per_pos_end)];
        k += m;
        done = true;
        f->stat_periodic_steps += m;
        out->periodic = true;
    }
    if (done) {
        finish_host_state();
        out->host_known = true; out->count = k; out->end = st; out->drift = drift; out->ksteps = ksteps;
        if (prof) std::fprintf(stderr, "[mrhip] schedule by the host: %lld outputs in %.3f ms%s\n", static_cast<long long>(k), (now() - t_begin) * 1e3, out->periodic ? " (periodic)" : "");
        return MRHIP_OK;
    }
    // ---- TABLES ---------------------------------------------------------------------------------------------
    if (!c.ok) return fail(MRHIP_ERR_INVALID_ARG, "device schedule requested for a rate it does not cover (internal)");
    const SchedPieceState hs{st.acc, st.xIdx, drift, ksteps};
    return enqueue_tables(f, x_len, est, y_capacity, count_out, k, ksteps, &hs, false, s, out);
}

// Wait for the result of a pending schedule (the FINISH kernel's mirror) and take it over.  A piece that did not verify
// is redone here with the serial loop and the tables continue behind it: *relaunch is set, the schedule is pending again
// and the caller enqueues the filter kernel once more (the first one found no outputs).
int sched_collect(mrhip_filter *f, int64_t x_len, int64_t est, int64_t y_capacity, long long *count_out, hipStream_t s, SchedOut *io, bool *relaunch)
{
    const SchedPlan &c = f->splan;
    *relaunch = false;
    if (!io->pending) return MRHIP_OK;
    MRHIP_CHECK_HIP(hipEventSynchronize(f->ev_rec));
    const DevStream r = *f->h_rec;
    const int64_t np = static_cast<int64_t>(io->pk0.size());
    if (r.sched_fail != kSchedNoFail) {
        // piece p did not verify: the pieces before it did, so the state the kernel left is the truth there
        const int64_t p = r.sched_fail;
        const SchedPieceState ps = *fail_state_host(f);
        f->stat_device_pieces += p + 1;
        f->stat_fallback_pieces += 1;
        if (int rc = wait_pinned_free(f)) return rc;
        ArbState hst{ps.acc, 0, 0.0, ps.xIdx, 0};
        const double acc_start = hst.acc;
        bool done = false;
        const int64_t cnt = run_arbitrary_schedule_piece(hst, f->delta, f->Nphi, x_len, static_cast<int32_t *>(f->pin_n), static_cast<double *>(f->pin_acc),
                                                         io->psteps[static_cast<size_t>(p)], &done);
        if (int rc = upload_entries(f, io->buf, io->pk0[static_cast<size_t>(p)], cnt, s)) return rc;
        double drift = ps.drift, ksteps = ps.ksteps;
        if (!done) {
            drift += wrap_half(hst.acc - sched_anchor_host(c, acc_start, static_cast<double>(cnt)), c.N);
            ksteps += static_cast<double>(cnt);
        }
        f->stat_host_steps += cnt;
        const int64_t k = io->pk0[static_cast<size_t>(p)] + cnt;
        *relaunch = true;
        io->pending = false;
        if (done) {
            hst.phiIdx = static_cast<int64_t>(std::floor(hst.acc));
            hst.alpha = hst.acc - static_cast<double>(hst.phiIdx);
            hst.inputDeficit = hst.xIdx - x_len;
            io->host_known = true; io->count = k; io->end = hst; io->drift = drift; io->ksteps = ksteps;
            return MRHIP_OK;
        }
        const SchedPieceState hs{hst.acc, hst.xIdx, drift, ksteps};
        return enqueue_tables(f, x_len, est, y_capacity, count_out, k, ksteps, &hs, false, s, io);
    }
    if (r.error == MRHIP_ERR_INVALID_ARG) return fail(MRHIP_ERR_INVALID_ARG, "phase schedule longer than the output-length bound (internal)");
    // (the pieces that were enqueued count as evaluated on the device: the last one holds the call's end)
    int64_t pl = 0;
    while (pl < np && io->pk0[static_cast<size_t>(pl)] + io->psteps[static_cast<size_t>(pl)] <= r.n_written) ++pl;
    f->stat_device_pieces += std::min<int64_t>(pl + 1, np);
    ArbState st{};
    st.acc = r.acc;
    st.phiIdx = static_cast<int64_t>(std::floor(r.acc));
    st.alpha = r.acc - static_cast<double>(st.phiIdx);
    st.inputDeficit = r.inputDeficit;
    st.xIdx = r.inputDeficit + x_len;
    io->count = r.n_written;
    io->end = st;
    io->drift = r.drift; io->ksteps = r.ksteps;
    io->per_pos_end = r.per_pos;
    io->pending = false;
    return MRHIP_OK;
}

}  // namespace mrhip
