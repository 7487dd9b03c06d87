// stream_state.hip -- the stream state ON THE DEVICE (DevStream / DevCall, mrhip_internal.h).
//
// Reference: every filt! ends by mutating the kernel object -- 𝜙Idx / inputDeficit (src/Filters.jl:571-572 Rational,
// :627-628 Decimator), 𝜙Accumulator / 𝜙Idx / α / inputDeficit (:731-735 Arbitrary, :836-838 Farrow, update() :663-673).
// Rounds 1-3 kept that state in the host object only, so a HIP graph could capture nothing but calls that leave it
// unchanged, and every FIRArbitrary / FIRFarrow call stalled the host in its middle for the output count.  Now the
// record lives in device memory and is kept current in stream order by every call:
//   * a call the host planned carries the end state it computed in its kernel arguments (one lane of the pair kernels
//     writes it: no extra launch on the streaming path), or pushes it with stream_set_kernel;
//   * a DEVICE-PLANNED call runs poly_plan_kernel (rational family, closed form of Filters.jl:558-571) or the schedule's
//     begin / finish kernels (kernels_schedule.hip) in front of its filter kernel: they read the record, leave what the
//     filter kernel needs in the DevCall, advance the record and mirror it into pinned host memory.
// Nothing here has a CPU fallback.
#include <cstring>

#include "mrhip_filter.h"

namespace mrhip {
namespace {

struct SetArgs {
    DevStream *rec, *mirror;
    DevCall *call;
    DevStream v;
    long long call_n_out;         // >= 0: arm the DevCall with this output count
    unsigned *zero;               // != NULL: kCounterBytes of hand-out counters to zero (reset(): no memset launch of its own)
};

__global__ __launch_bounds__(64) void stream_set_kernel(SetArgs a)
{
    if (a.zero)
        for (unsigned i = threadIdx.x; i < kCounterBytes / 4; i += 64) a.zero[i] = 0u;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DevStream v = a.v;
    v.calls = a.rec->calls + (a.v.calls != 0 ? 1 : 0);        // (v.calls is a flag here: this push completes a call)
    v.fallback_steps = a.rec->fallback_steps;
    if (a.v.error < 0) v.error = a.rec->error;                // error < 0: keep a pending asynchronous error
    *a.rec = v;
    *a.mirror = v;
    if (a.call_n_out >= 0) { DevCall c{}; c.n_out = a.call_n_out; *a.call = c; }
}

struct PlanArgs {
    DevStream *rec, *mirror;
    DevCall *call;
    long long *count_out;
    long long L, M, x_len, P, y_capacity;
    int kind, nch;
    const DevCall *x_from;        // a chained call: the input length is the count this record holds (x_len above: its upper bound)
};

// filt! of the rational family planned where the state lives (Filters.jl:543-547 short input, :558-571 the loop in closed
// form): one lane.
__global__ __launch_bounds__(64) void poly_plan_kernel(PlanArgs a)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DevStream r = *a.rec;
    long long x_len = a.x_len;
    if (a.x_from) {
        x_len = a.x_from->n_out;
        if (x_len > a.x_len) { x_len = a.x_len; r.error = MRHIP_ERR_BUFFER_TOO_SMALL; }   // cannot happen: the bound is the previous stage's own
        if (x_len < 0) x_len = 0;
    }
    const CallPlanPOD p = plan_rational_pod(a.kind, a.L, a.M, r.phiIdx, r.inputDeficit, x_len);
    DevCall c{};
    c.x_len = x_len;
    c.n_out = p.n_out;
    if (c.n_out > a.y_capacity) {           // cannot happen: the host checked the capacity against the largest count of any state
        c.n_out = a.y_capacity;
        r.error = MRHIP_ERR_BUFFER_TOO_SMALL;
    }
    c.u0 = p.phi0 - 1;
    c.d0 = p.d0;
    const long long spc = (c.n_out + a.P - 1) / a.P;
    c.steps_per_channel = static_cast<unsigned>(spc);
    c.total_steps = static_cast<unsigned>(spc * a.nch);
    c.spc_magic = spc <= 1 ? 0xffffffffu : static_cast<unsigned>((1ULL << 32) / static_cast<unsigned long long>(spc));
    *a.call = c;
    if (a.kind == MRHIP_FIR_DECIMATOR || a.kind == MRHIP_FIR_RATIONAL) {
        r.phiIdx = a.kind == MRHIP_FIR_DECIMATOR ? r.phiIdx : p.phi_end;
        r.inputDeficit = p.d_end;
    }
    r.n_written = c.n_out;
    r.calls += 1;
    *a.rec = r;
    *a.mirror = r;
    if (a.count_out) *a.count_out = c.n_out;
}

}  // namespace

int rec_alloc(mrhip_filter *f)
{
    void *d = nullptr;
    MRHIP_CHECK_HIP(hipMalloc(&d, 256 + 2 * 256));
    f->d_rec = static_cast<DevStream *>(d);
    static_assert(sizeof(DevStream) <= 256 && sizeof(DevCall) <= 256, "record slots");
    for (int b = 0; b < 2; ++b) f->d_calls[b] = reinterpret_cast<DevCall *>(static_cast<unsigned char *>(d) + 256 * (b + 1));
    f->d_call = f->d_calls[0];
    void *h = nullptr;
    MRHIP_CHECK_HIP(hipHostMalloc(&h, 512, hipHostMallocMapped));
    std::memset(h, 0, 512);
    f->h_rec = static_cast<DevStream *>(h);
    MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->ev_rec, hipEventDisableTiming));
    if (f->kind == MRHIP_FIR_ARBITRARY || f->kind == MRHIP_FIR_FARROW) {      // the schedule's own stream (mrhip_filter.h: s_sched)
        // LOWEST priority: the schedule of call i+1 and the filter kernel of call i become runnable at the same moment (both wait
        // for the filter kernel of call i-1); the filter kernel's persistent workgroups must be placed first and the schedule's fill
        // what is left, not the other way round (profiles/r04/experiments.md S).  MRHIP_SCHED_PRIO=0: default priority.
        int prio_least = 0, prio_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_least = 0; }
        const char *pv = std::getenv("MRHIP_SCHED_PRIO");
        if (pv && *pv == '0') prio_least = 0;
        if (pv && *pv == 'h') prio_least = prio_greatest;      // ("high")
        MRHIP_CHECK_HIP(hipStreamCreateWithPriority(&f->s_sched, hipStreamNonBlocking, prio_least));
        for (int b = 0; b < 2; ++b) {
            MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->ev_fin[b], hipEventDisableTiming));
            MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->ev_filt[b], hipEventDisableTiming));
        }
        MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->ev_sdirty, hipEventDisableTiming));
    }
    DevStream v{};
    v.phiIdx = 1; v.inputDeficit = 1; v.acc = 1.0;
    v.sched_fail = kSchedNoFail;
    MRHIP_CHECK_HIP(hipMemsetAsync(d, 0, 256 + 2 * 256, f->own_stream));
    MRHIP_CHECK_HIP(hipMemcpyAsync(f->d_rec, &v, sizeof v, hipMemcpyHostToDevice, f->own_stream));
    MRHIP_CHECK_HIP(hipStreamSynchronize(f->own_stream));
    *f->h_rec = v;
    return MRHIP_OK;
}

void rec_free(mrhip_filter *f)
{
    if (f->s_sched) { (void)hipStreamSynchronize(f->s_sched); (void)hipStreamDestroy(f->s_sched); }
    for (hipEvent_t e : {f->ev_fin[0], f->ev_fin[1], f->ev_filt[0], f->ev_filt[1]})
        if (e) (void)hipEventDestroy(e);
    if (f->d_rec) (void)hipFree(f->d_rec);
    if (f->h_rec) (void)hipHostFree(f->h_rec);
    if (f->ev_rec) (void)hipEventDestroy(f->ev_rec);
    f->d_rec = nullptr; f->d_call = nullptr; f->h_rec = nullptr; f->ev_rec = nullptr; f->s_sched = nullptr;
}

// the pinned mirror as the device sees it
static DevStream *mirror_dev(mrhip_filter *f)
{
    void *p = nullptr;
    if (hipHostGetDevicePointer(&p, f->h_rec, 0) != hipSuccess) { (void)hipGetLastError(); return f->h_rec; }
    return static_cast<DevStream *>(p);
}

int rec_push(mrhip_filter *f, hipStream_t s, long long call_n_out, long long n_written, unsigned *zero_counters)
{
    SetArgs a{};
    a.zero = zero_counters;
    a.rec = f->d_rec; a.mirror = mirror_dev(f); a.call = f->d_call;
    a.v.phiIdx = f->phiIdx; a.v.inputDeficit = f->inputDeficit; a.v.acc = f->phiAcc;
    a.v.drift = f->sched_drift; a.v.ksteps = f->sched_ksteps; a.v.per_pos = f->per_pos;
    a.v.n_written = n_written >= 0 ? n_written : 0;
    a.v.calls = n_written >= 0 ? 1 : 0;
    a.v.error = n_written >= 0 ? -1 : 0;      // a completed call keeps a pending error; reset / set_state clear it
    a.v.sched_fail = kSchedNoFail;
    a.call_n_out = call_n_out;
    hipLaunchKernelGGL(stream_set_kernel, dim3(1), dim3(64), 0, s, a);
    MRHIP_CHECK_HIP(hipGetLastError());
    return MRHIP_OK;
}

int rec_pull(mrhip_filter *f)
{
    // graph replays run on streams the library never saw: after a capture only the whole device is a safe wait
    if (f->captured) {
        if (any_capture_active())
            return fail(MRHIP_ERR_UNSUPPORTED, "a stream is being captured: the host-side state of a filter whose calls were captured once needs a device-wide wait (its "
                                               "replays run on streams the library never saw), which would invalidate that capture -- ask before the capture begins or after it ends");
        MRHIP_CHECK_HIP(device_sync_relaxed());
    }
    else {
        if (f->last_stream_valid && hipStreamSynchronize(f->last_stream) != hipSuccess) {
            (void)hipGetLastError();
            MRHIP_CHECK_HIP(device_sync_relaxed());
        }
        if (f->s_sched) MRHIP_CHECK_HIP(hipStreamSynchronize(f->s_sched));
    }
    f->async_pending = false;
    const DevStream r = *f->h_rec;
    f->phiIdx = r.phiIdx; f->inputDeficit = r.inputDeficit;
    if (f->kind == MRHIP_FIR_ARBITRARY || f->kind == MRHIP_FIR_FARROW) {
        f->phiAcc = r.acc;
        f->phiIdx = static_cast<int64_t>(__builtin_floor(r.acc));      // Filters.jl:671-672
        f->alpha = r.acc - static_cast<double>(f->phiIdx);
        f->xIdx = r.inputDeficit;
        f->sched_drift = r.drift; f->sched_ksteps = r.ksteps;
        f->per_pos = r.per_pos;
    }
    f->mirror_valid = true;
    return MRHIP_OK;
}

hipError_t launch_poly_plan(mrhip_filter *f, int64_t x_len, long long P, long long y_capacity, long long *count_out, hipStream_t s, const DevCall *x_from)
{
    PlanArgs a{};
    a.x_from = x_from;
    a.rec = f->d_rec; a.mirror = mirror_dev(f); a.call = f->d_call; a.count_out = count_out;
    a.L = f->L; a.M = f->M; a.x_len = x_len; a.P = P > 0 ? P : 1; a.y_capacity = y_capacity;
    a.kind = f->kind; a.nch = static_cast<int>(f->nch);
    hipLaunchKernelGGL(poly_plan_kernel, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace mrhip
