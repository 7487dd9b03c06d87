// sharded.cpp -- mrhip_sharded_* (include/multirate_hip.h): one FIRFilter whose channels are split over several GPUs, behind the C ABI.
//
// The reference's seam is filt(self, x) on one Vector (src/Filters.jl:577-587); a multi-channel signal is one FIRFilter per channel, and
// channels are fully independent (each owns its history; only the read-only taps are shared) -- SURVEY.md 8(e).  A sharded filter is
// therefore n ordinary filters, shard i on device[i] with a contiguous range of the channels (the split of sharding.py: the first
// nchannels % n shards take one channel more); there is NO exchange during compute.  One process drives all devices: every shard has a
// stream of its own on its device, calls on different shards overlap, host data goes through one thread per shard.  The only data
// movement between devices is the final gather of the outputs (BASELINE.json config 5), peer copies over xGMI on the shards' streams.
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "mrhip_filter.h"

struct mrhip_sharded {
    std::vector<mrhip_filter *> shard;     // NULL for a shard without channels
    std::vector<int> device;
    std::vector<int64_t> start, count;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> ev;            // orders a shard's stream against a caller's stream (mrhip_sharded_wait_stream / _signal_stream)
    int64_t nch = 0;
    int tx = 0, ty = 0;
};

using namespace mrhip;

namespace {
struct DevGuard {
    int prev = -1;
    explicit DevGuard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != d) (void)hipSetDevice(d); }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" {

int mrhip_sharded_create(int ctor, const void *h, int64_t hLen, int tap_dtype, int64_t num, int64_t den, double rate, int64_t Nphi,
                         int64_t polyorder, int sample_dtype, int64_t nchannels, const int *devices, int ndevices, mrhip_sharded **out)
{
    if (!out) return fail(MRHIP_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!devices || ndevices < 1) return fail(MRHIP_ERR_INVALID_ARG, "a sharded filter takes one or more devices");
    if (nchannels < 1) return fail(MRHIP_ERR_INVALID_ARG, "nchannels must be >= 1");
    if (ctor < 0 || ctor > 2) return fail(MRHIP_ERR_INVALID_ARG, "ctor: 0 = FIRFilter(h, ratio), 1 = FIRFilter(h, rate, Nphi), 2 = FIRFilter(h, rate, Nphi, polyorder)");
    auto *s = new mrhip_sharded();
    s->nch = nchannels; s->tx = sample_dtype; s->ty = mrhip_output_dtype(tap_dtype, sample_dtype);
    const int64_t base = nchannels / ndevices, extra = nchannels % ndevices;
    for (int i = 0; i < ndevices; ++i) {
        const int64_t cnt = base + (i < extra ? 1 : 0);
        s->device.push_back(devices[i]);
        s->start.push_back(i * base + std::min<int64_t>(i, extra));
        s->count.push_back(cnt);
        s->shard.push_back(nullptr);
        s->stream.push_back(nullptr);
        s->ev.push_back(nullptr);
        if (cnt == 0) continue;
        mrhip_filter *f = nullptr;
        int rc = ctor == 0 ? mrhip_create_rational(h, hLen, tap_dtype, num, den, sample_dtype, cnt, devices[i], &f)
               : ctor == 1 ? mrhip_create_arbitrary(h, hLen, tap_dtype, rate, Nphi, sample_dtype, cnt, devices[i], &f)
                           : mrhip_create_farrow(h, hLen, tap_dtype, rate, Nphi, polyorder, sample_dtype, cnt, devices[i], &f);
        if (rc == MRHIP_OK) {
            DevGuard g(devices[i]);
            if (hipStreamCreateWithFlags(&s->stream[static_cast<size_t>(i)], hipStreamNonBlocking) != hipSuccess) rc = fail(MRHIP_ERR_HIP, "hipStreamCreate failed");
            else if (hipEventCreateWithFlags(&s->ev[static_cast<size_t>(i)], hipEventDisableTiming) != hipSuccess) rc = fail(MRHIP_ERR_HIP, "hipEventCreate failed");
        }
        if (rc != MRHIP_OK) {
            const std::string msg = mrhip_last_error();
            if (f) mrhip_destroy(f);
            mrhip_sharded_destroy(s);
            return fail(rc, msg);
        }
        s->shard[static_cast<size_t>(i)] = f;
    }
    *out = s;
    return MRHIP_OK;
}

void mrhip_sharded_destroy(mrhip_sharded *s)
{
    if (!s) return;
    for (size_t i = 0; i < s->shard.size(); ++i) {
        // (the shard's kernels run on its stream: they are through before the filter they read is freed)
        if (s->stream[i]) { DevGuard g(s->device[i]); (void)hipStreamSynchronize(s->stream[i]); }
        if (s->shard[i]) mrhip_destroy(s->shard[i]);
        if (s->ev[i]) { DevGuard g(s->device[i]); (void)hipEventDestroy(s->ev[i]); }
        if (s->stream[i]) { DevGuard g(s->device[i]); (void)hipStreamDestroy(s->stream[i]); }
    }
    delete s;
}

int mrhip_sharded_nshards(const mrhip_sharded *s) { return s ? static_cast<int>(s->shard.size()) : -1; }

int mrhip_sharded_shard(const mrhip_sharded *s, int i, int64_t *start, int64_t *count, int *device, mrhip_filter **filter)
{
    if (!s || i < 0 || i >= static_cast<int>(s->shard.size())) return fail(MRHIP_ERR_INVALID_ARG, "no such shard");
    if (start) *start = s->start[static_cast<size_t>(i)];
    if (count) *count = s->count[static_cast<size_t>(i)];
    if (device) *device = s->device[static_cast<size_t>(i)];
    if (filter) *filter = s->shard[static_cast<size_t>(i)];
    return MRHIP_OK;
}

int64_t mrhip_sharded_next_output_count(const mrhip_sharded *s, int64_t inputlength)
{
    if (!s) return -1;
    for (mrhip_filter *f : s->shard)
        if (f) return mrhip_next_output_count(f, inputlength);     // the state machine is data independent: every shard agrees
    return 0;
}

int64_t mrhip_sharded_outputlength(const mrhip_sharded *s, int64_t inputlength)
{
    if (!s) return -1;
    for (mrhip_filter *f : s->shard)
        if (f) return mrhip_outputlength(f, inputlength);
    return 0;
}

int mrhip_sharded_reset(mrhip_sharded *s)
{
    if (!s) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    for (mrhip_filter *f : s->shard)
        if (f) if (int rc = mrhip_reset(f)) return rc;
    return MRHIP_OK;
}

int mrhip_sharded_filt_device(mrhip_sharded *s, const void *const *x, int64_t x_len, const int64_t *x_stride, void *const *y, int64_t y_capacity,
                              const int64_t *y_stride, int64_t *n_written)
{
    if (!s || !x || !y) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    if (n_written) *n_written = 0;
    // reference: error() before any work (Filters.jl:460,503,550) -- on EVERY shard before anything is enqueued on any
    const int64_t want = mrhip_sharded_next_output_count(s, x_len);
    const bool estimate = !s->shard.empty() && [&] { for (mrhip_filter *f : s->shard) if (f) return f->kind == MRHIP_FIR_ARBITRARY || f->kind == MRHIP_FIR_FARROW; return false; }();
    if (!estimate && want > y_capacity) return fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small");
    // every shard's arguments before anything is enqueued on any: a call that fails half way leaves the shards at different stream
    // positions (FIRArbitrary / FIRFarrow: the count is an estimate, the room is checked against the reference's outputlength)
    for (size_t i = 0; i < s->shard.size(); ++i) {
        mrhip_filter *f = s->shard[i];
        if (!f) continue;
        const int64_t xs = x_stride ? x_stride[i] : x_len, ys = y_stride ? y_stride[i] : y_capacity;
        if (x_len < 0 || y_capacity < 0) return fail(MRHIP_ERR_INVALID_ARG, "negative length");
        if (x_len > 0 && (!x[i] || !y[i])) return fail(MRHIP_ERR_INVALID_ARG, "a shard's x or y is NULL");
        if (s->count[i] > 1 && (xs < x_len || ys < std::min<int64_t>(want, y_capacity))) return fail(MRHIP_ERR_INVALID_ARG, "a shard's stride is shorter than its rows");
        if (estimate && mrhip_outputlength(f, x_len) > y_capacity) return fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small");
    }
    int64_t got_all = -1;
    for (size_t i = 0; i < s->shard.size(); ++i) {
        mrhip_filter *f = s->shard[i];
        if (!f) continue;
        int64_t got = 0;
        const int64_t xs = x_stride ? x_stride[i] : x_len, ys = y_stride ? y_stride[i] : y_capacity;
        if (int rc = mrhip_filt_device(f, x[i], x_len, xs, y[i], y_capacity, ys, &got, s->stream[i])) return rc;
        if (got_all >= 0 && got != got_all) return fail(MRHIP_ERR_HIP, "shards disagree on the output count (internal)");
        got_all = got;
    }
    if (n_written) *n_written = std::max<int64_t>(got_all, 0);
    return MRHIP_OK;
}

int mrhip_sharded_filt_host(mrhip_sharded *s, const void *x, int64_t x_len, int64_t x_stride, void *y, int64_t y_capacity, int64_t y_stride,
                            int64_t *n_written)
{
    if (!s) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    if (n_written) *n_written = 0;
    if (x_len > 0 && (!x || !y)) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    if (s->nch > 1 && x_stride < x_len) return fail(MRHIP_ERR_INVALID_ARG, "x_stride < x_len");
    const size_t xe = dtype_size(s->tx), ye = dtype_size(s->ty);
    const size_t n = s->shard.size();
    std::vector<int> rcs(n, MRHIP_OK);
    std::vector<int64_t> got(n, 0);
    std::vector<std::string> errs(n);
    // one host thread per shard: the copies and kernels of different devices overlap (mrhip_filt_host itself pipelines within a shard)
    std::vector<std::thread> th;
    for (size_t i = 0; i < n; ++i) {
        if (!s->shard[i]) continue;
        th.emplace_back([&, i] {
            const unsigned char *xi = static_cast<const unsigned char *>(x) + static_cast<size_t>(s->start[i]) * static_cast<size_t>(x_stride) * xe;
            unsigned char *yi = static_cast<unsigned char *>(y) + static_cast<size_t>(s->start[i]) * static_cast<size_t>(y_stride) * ye;
            rcs[i] = mrhip_filt_host(s->shard[i], xi, x_len, x_stride, yi, y_capacity, y_stride, &got[i]);
            if (rcs[i]) errs[i] = mrhip_last_error();        // (the message is thread-local: carry it over)
        });
    }
    for (auto &t : th) t.join();
    int64_t got_all = -1;
    for (size_t i = 0; i < n; ++i) {
        if (!s->shard[i]) continue;
        if (rcs[i]) return fail(rcs[i], errs[i]);
        if (got_all >= 0 && got[i] != got_all) return fail(MRHIP_ERR_HIP, "shards disagree on the output count (internal)");
        got_all = got[i];
    }
    if (n_written) *n_written = std::max<int64_t>(got_all, 0);
    return MRHIP_OK;
}

int mrhip_sharded_gather(mrhip_sharded *s, const void *const *y, int64_t n_out, const int64_t *y_stride, void *dst, int64_t dst_stride, int dst_device)
{
    if (!s || !y || (n_out > 0 && !dst)) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    if (n_out < 0 || dst_stride < n_out) return fail(MRHIP_ERR_INVALID_ARG, "dst_stride < n_out");
    const size_t ye = dtype_size(s->ty);
    for (size_t i = 0; i < s->shard.size(); ++i) {
        if (!s->shard[i] || s->count[i] == 0 || n_out == 0) continue;
        DevGuard g(s->device[i]);
        const int64_t ys = y_stride ? y_stride[i] : n_out;
        unsigned char *d = static_cast<unsigned char *>(dst) + static_cast<size_t>(s->start[i]) * static_cast<size_t>(dst_stride) * ye;
        // rows [start, start + count) of the result, in stream order behind the shard's filter kernel (peer copy over xGMI when the
        // result lives on another device; unified addressing names the source and destination devices)
        if (ys == n_out && dst_stride == n_out) {
            MRHIP_CHECK_HIP(hipMemcpyPeerAsync(d, dst_device, y[i], s->device[i], static_cast<size_t>(s->count[i]) * static_cast<size_t>(n_out) * ye, s->stream[i]));
        } else {
            MRHIP_CHECK_HIP(hipMemcpy2DAsync(d, static_cast<size_t>(dst_stride) * ye, y[i], static_cast<size_t>(ys) * ye, static_cast<size_t>(n_out) * ye,
                                             static_cast<size_t>(s->count[i]), hipMemcpyDefault, s->stream[i]));
        }
    }
    return MRHIP_OK;
}

// shard i's stream behind what `stream` (a stream of the shard's device; NULL: the default stream) holds so far: inputs a caller's
// kernels are still writing, buffers its allocator has handed out in stream order
int mrhip_sharded_wait_stream(mrhip_sharded *s, int i, void *stream)
{
    if (!s || i < 0 || i >= static_cast<int>(s->shard.size())) return fail(MRHIP_ERR_INVALID_ARG, "no such shard");
    if (!s->stream[static_cast<size_t>(i)]) return MRHIP_OK;
    DevGuard g(s->device[static_cast<size_t>(i)]);
    MRHIP_CHECK_HIP(hipEventRecord(s->ev[static_cast<size_t>(i)], static_cast<hipStream_t>(stream)));
    MRHIP_CHECK_HIP(hipStreamWaitEvent(s->stream[static_cast<size_t>(i)], s->ev[static_cast<size_t>(i)], 0));
    return MRHIP_OK;
}

// ... and `stream` behind what shard i's stream holds so far: the caller's later kernels see the shard's outputs, its allocator may
// re-use their memory in stream order
int mrhip_sharded_signal_stream(mrhip_sharded *s, int i, void *stream)
{
    if (!s || i < 0 || i >= static_cast<int>(s->shard.size())) return fail(MRHIP_ERR_INVALID_ARG, "no such shard");
    if (!s->stream[static_cast<size_t>(i)]) return MRHIP_OK;
    DevGuard g(s->device[static_cast<size_t>(i)]);
    MRHIP_CHECK_HIP(hipEventRecord(s->ev[static_cast<size_t>(i)], s->stream[static_cast<size_t>(i)]));
    MRHIP_CHECK_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), s->ev[static_cast<size_t>(i)], 0));
    return MRHIP_OK;
}

int mrhip_sharded_synchronize(mrhip_sharded *s)
{
    if (!s) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    for (size_t i = 0; i < s->shard.size(); ++i) {
        if (!s->stream[i]) continue;
        DevGuard g(s->device[i]);
        MRHIP_CHECK_HIP(hipStreamSynchronize(s->stream[i]));
    }
    return MRHIP_OK;
}

}  // extern "C"
