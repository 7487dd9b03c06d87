// cascade.cpp -- mrhip_cascade_* (include/multirate_hip.h): a chain of stateful filters run back to back on one
// stream with the intermediate signals resident in HBM, and mrhip_arbitrary_tapsforphase.
//
// The reference has no cascade object: a user chains filt(f2, filt(f1, x)) by hand and every intermediate is a
// fresh host Vector (src/Filters.jl:475-873).  Here every stage's output count is known on the host before anything
// runs (the state machines are data independent), so the chain is enqueued without reading anything back.
#include <algorithm>
#include <cmath>
#include <vector>

#include "mrhip_filter.h"

struct mrhip_cascade {
    std::vector<mrhip_filter *> stages;
    void *buf[2] = {nullptr, nullptr};     // ping-pong intermediates, [nch][cap] planar
    size_t cap_bytes[2] = {0, 0};
    int device = 0;
    int64_t nch = 1;
};

using namespace mrhip;

namespace {

int grow(mrhip_cascade *c, int slot, size_t bytes)
{
    if (bytes <= c->cap_bytes[slot]) return MRHIP_OK;
    if (c->buf[slot]) MRHIP_CHECK_HIP(hipFree(c->buf[slot]));     // hipFree waits for the device: earlier users are done
    c->buf[slot] = nullptr; c->cap_bytes[slot] = 0;
    const size_t want = bytes + bytes / 8;
    MRHIP_CHECK_HIP(hipMalloc(&c->buf[slot], want));
    c->cap_bytes[slot] = want;
    return MRHIP_OK;
}

}  // namespace

extern "C" {

int mrhip_cascade_create(mrhip_filter *const *stages, int nstages, mrhip_cascade **out)
{
    if (!out) return fail(MRHIP_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!stages || nstages < 1) return fail(MRHIP_ERR_INVALID_ARG, "a cascade takes one or more stages");
    for (int i = 0; i < nstages; ++i) {
        if (!stages[i]) return fail(MRHIP_ERR_INVALID_ARG, "NULL stage");
        if (stages[i]->nch != stages[0]->nch || stages[i]->device != stages[0]->device)
            return fail(MRHIP_ERR_INVALID_ARG, "all stages of a cascade must have the same nchannels and device");
        if (i > 0 && stages[i]->tx != stages[i - 1]->ty)
            return fail(MRHIP_ERR_INVALID_ARG, "a stage's sample dtype must be the previous stage's output dtype");
    }
    auto *c = new mrhip_cascade();
    c->stages.assign(stages, stages + nstages);
    c->device = stages[0]->device;
    c->nch = stages[0]->nch;
    *out = c;
    return MRHIP_OK;
}

void mrhip_cascade_destroy(mrhip_cascade *c)
{
    if (!c) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(c->device);
    for (void *p : c->buf)
        if (p) (void)hipFree(p);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete c;
}

int64_t mrhip_cascade_outputlength(const mrhip_cascade *c, int64_t n)
{
    if (!c) return -1;
    for (const mrhip_filter *f : c->stages) n = std::max<int64_t>(mrhip_outputlength(f, n), 0);
    return n;
}

int64_t mrhip_cascade_next_output_count(const mrhip_cascade *c, int64_t n)
{
    if (!c || n < 0) return -1;
    for (const mrhip_filter *f : c->stages) n = n > 0 ? std::max<int64_t>(mrhip_next_output_count(f, n), 0) : 0;
    return n;
}

int mrhip_cascade_filt_device(mrhip_cascade *c, const void *x, int64_t x_len, int64_t x_stride, void *y, int64_t y_capacity,
                              int64_t y_stride, int64_t *n_written, void *stream)
{
    if (!c) return fail(MRHIP_ERR_INVALID_ARG, "NULL cascade");
    if (n_written) *n_written = 0;
    if (x_len < 0 || y_capacity < 0) return fail(MRHIP_ERR_INVALID_ARG, "negative length");
    const size_t ns = c->stages.size();
    // Under HIP-graph capture every replay runs the stages with the lengths planned NOW (a stage's input length is baked into
    // the next stage's launch): each stage must therefore turn the same number of samples into the same number of outputs on
    // every replay -- rational family with inputlength * L a multiple of M (the state then returns to itself; FIRStandard and
    // FIRInterpolator always do).  Anything else is refused instead of replaying stale lengths.  A captured stage is planned
    // on the device and wants room for mrhip_outputlength_bound outputs: the buffers between the stages get that.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
    if (capturing) {
        int64_t m = x_len;
        for (size_t i = 0; i < ns; ++i) {
            const mrhip_filter *f = c->stages[i];
            if (f->kind == MRHIP_FIR_ARBITRARY || f->kind == MRHIP_FIR_FARROW)
                return fail(MRHIP_ERR_UNSUPPORTED, "cascade under graph capture: a FIRArbitrary / FIRFarrow stage produces a count that differs from replay to replay; capture its calls on their own (mrhip_filt_device_async)");
            if (static_cast<__int128>(m) * f->L % f->M != 0)
                return fail(MRHIP_ERR_UNSUPPORTED, "cascade under graph capture: a stage's inputlength * L is not a multiple of M, so its output count differs from replay to replay");
            m = static_cast<int64_t>(static_cast<__int128>(m) * f->L / f->M);
        }
    }
    // counts of every stage first (pure functions of state and length): the error check precedes any work
    std::vector<int64_t> cnt(ns), room(ns);
    int64_t n = x_len;
    for (size_t i = 0; i < ns; ++i) {
        const int64_t in_n = n;
        if (capturing) n = static_cast<int64_t>(static_cast<__int128>(n) * c->stages[i]->L / c->stages[i]->M);   // (exact whatever the state: checked above; the host's view of the state may not be read during a capture)
        else n = n > 0 ? std::max<int64_t>(mrhip_next_output_count(c->stages[i], n), 0) : 0;
        cnt[i] = n;
        room[i] = std::max<int64_t>(mrhip_outputlength_bound(c->stages[i], in_n), n);   // (what a device-planned call of this length wants: a plain call of the size warms the buffers up for a capture)
    }
    if ((capturing ? room[ns - 1] : n) > y_capacity) return fail(MRHIP_ERR_BUFFER_TOO_SMALL, capturing ? "buffer is too small: a captured call needs room for mrhip_outputlength_bound outputs of the last stage" : "buffer is too small");
    if (n > 0 && !y) return fail(MRHIP_ERR_INVALID_ARG, "y is NULL");
    if (c->nch > 1 && n > 0 && y_stride < n) return fail(MRHIP_ERR_INVALID_ARG, "y_stride < output count");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != c->device && hipSetDevice(c->device) != hipSuccess) return fail(MRHIP_ERR_HIP, "hipSetDevice failed");
    int rc = MRHIP_OK;
    for (size_t i = 0; i + 1 < ns && !rc; ++i) {
        const size_t want = std::max<size_t>(static_cast<size_t>(room[i]) * dtype_size(c->stages[i]->ty) * static_cast<size_t>(c->nch), 16);
        if (capturing && want > c->cap_bytes[i & 1])
            rc = fail(MRHIP_ERR_UNSUPPORTED, "cascade under graph capture: the buffers between the stages are allocated by the first call of a size; run one plain call of this size before capturing");
    }
    for (size_t i = 0; i + 1 < ns && !rc; ++i)
        rc = grow(c, static_cast<int>(i & 1), std::max<size_t>(static_cast<size_t>(room[i]) * dtype_size(c->stages[i]->ty) * static_cast<size_t>(c->nch), 16));
    const void *in = x;
    int64_t in_len = x_len, in_stride = x_stride;
    for (size_t i = 0; i < ns && !rc && in_len > 0; ++i) {
        const bool last = i + 1 == ns;
        void *outp = last ? y : c->buf[i & 1];
        const int64_t cap = last ? y_capacity : room[i];
        const int64_t stride = last ? y_stride : std::max<int64_t>(room[i], 1);
        int64_t got = 0;
        rc = mrhip_filt_device(c->stages[i], in, in_len, in_stride, outp, cap, stride, &got, stream);
        if (!rc && got != cnt[i]) rc = fail(MRHIP_ERR_INVALID_ARG, "cascade: a stage produced a different count than planned");
        in = outp; in_len = got; in_stride = stride;
    }
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    if (!rc && n_written) *n_written = n;
    return rc;
}

int mrhip_cascade_filt_device_async(mrhip_cascade *c, const void *x, int64_t x_len, int64_t x_stride, void *y, int64_t y_capacity,
                                    int64_t y_stride, int64_t *count_out, void *stream)
{
    if (!c) return fail(MRHIP_ERR_INVALID_ARG, "NULL cascade");
    if (x_len < 0 || y_capacity < 0) return fail(MRHIP_ERR_INVALID_ARG, "negative length");
    const size_t ns = c->stages.size();
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
    if (x_len == 0) {                  // nothing to do: no stage runs, every state stays (and no stage's call record goes stale)
        if (count_out) MRHIP_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int64_t), static_cast<hipStream_t>(stream)));
        return MRHIP_OK;
    }
    // every stage is sized for the BOUND of what the stage before it can produce
    std::vector<int64_t> room(ns);
    int64_t n = x_len;
    for (size_t i = 0; i < ns; ++i) {
        const mrhip_filter *f = c->stages[i];
        n = std::max<int64_t>(mrhip_outputlength_bound(f, n), 0);
        room[i] = n;
    }
    if (room[ns - 1] > y_capacity) return fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small: an asynchronous cascade needs room for the bound of the last stage's output");
    if (room[ns - 1] > 0 && !y) return fail(MRHIP_ERR_INVALID_ARG, "y is NULL");
    if (c->nch > 1 && y_stride < room[ns - 1]) return fail(MRHIP_ERR_INVALID_ARG, "y_stride < output bound");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != c->device && hipSetDevice(c->device) != hipSuccess) return fail(MRHIP_ERR_HIP, "hipSetDevice failed");
    int rc = MRHIP_OK;
    for (size_t i = 0; i + 1 < ns && !rc; ++i) {
        const size_t want = std::max<size_t>(static_cast<size_t>(room[i]) * dtype_size(c->stages[i]->ty) * static_cast<size_t>(c->nch), 16);
        if (capturing && want > c->cap_bytes[i & 1])
            rc = fail(MRHIP_ERR_UNSUPPORTED, "cascade under graph capture: the buffers between the stages are allocated by the first call of a size; run one plain call of this size before capturing");
        if (!rc) rc = grow(c, static_cast<int>(i & 1), want);
    }
    const void *in = x;
    int64_t in_len = x_len, in_stride = x_stride;
    for (size_t i = 0; i < ns && !rc; ++i) {
        const bool last = i + 1 == ns;
        void *outp = last ? y : c->buf[i & 1];
        const int64_t cap = last ? y_capacity : room[i];
        const int64_t stride = last ? y_stride : std::max<int64_t>(room[i], 1);
        int64_t *cnt = last ? count_out : nullptr;
        rc = i == 0 ? mrhip_filt_device_async(c->stages[0], in, in_len, in_stride, outp, cap, stride, cnt, stream)
                    : mrhip_filt_device_chained(c->stages[i], c->stages[i - 1], in, in_len, in_stride, outp, cap, stride, cnt, stream);
        in = outp; in_len = room[i]; in_stride = stride;
    }
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    return rc;
}

int mrhip_cascade_reset(mrhip_cascade *c)
{
    if (!c) return fail(MRHIP_ERR_INVALID_ARG, "NULL cascade");
    for (mrhip_filter *f : c->stages)
        if (int rc = mrhip_reset(f)) return rc;
    return MRHIP_OK;
}

int mrhip_arbitrary_tapsforphase(const mrhip_filter *f, double phase, void *host_out)
{
    if (!f || !host_out) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    if (f->kind != MRHIP_FIR_ARBITRARY) return fail(MRHIP_ERR_INVALID_ARG, "not a FIRArbitrary filter");
    if (!(phase >= 0.0 && phase <= static_cast<double>(f->Nphi) + 1.0))
        return fail(MRHIP_ERR_INVALID_ARG, "phase must be >= 0 and <= Nphi+1");             // Filters.jl:678
    double ip = 0.0;
    const double alpha = std::modf(phase, &ip);                                                 // :681
    const int64_t col = static_cast<int64_t>(ip);                                               // :682
    if (col < 1 || col > f->Nphi) return fail(MRHIP_ERR_INVALID_ARG, "phase selects column 0 or Nphi+1 of the filter bank (BoundsError in the reference)");
    const size_t base = static_cast<size_t>(col - 1) * static_cast<size_t>(f->T);             // column-major T x Nphi
    for (int64_t i = 0; i < f->T; ++i) {                                                        // :684-686, Float64 arithmetic, stored as T
        if (f->th == MRHIP_F32) {
            const double p = static_cast<double>(reinterpret_cast<const float *>(f->h_taps.data())[base + i]);
            const double d = static_cast<double>(reinterpret_cast<const float *>(f->h_dtaps.data())[base + i]);
            const double t = alpha * d;
            static_cast<float *>(host_out)[i] = static_cast<float>(p + t);
        } else {
            const double p = reinterpret_cast<const double *>(f->h_taps.data())[base + i];
            const double d = reinterpret_cast<const double *>(f->h_dtaps.data())[base + i];
            const double t = alpha * d;
            static_cast<double *>(host_out)[i] = p + t;
        }
    }
    return MRHIP_OK;
}

}  // extern "C"
