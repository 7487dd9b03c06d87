// api.hip -- the extern "C" surface of libmultirate_hip.so (include/multirate_hip.h).
//
// There is deliberately NO CPU execution path in this library: if no gfx950 device is visible
// every constructor fails with MRHIP_ERR_NO_DEVICE.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <unordered_map>

#include "mrhip_filter.h"

namespace mrhip {

// kernels_farrow_wave.hip: FIRFarrow for fewer than four channels (one lane per output, windows straight from global memory)
bool plan_farrow_wave(const FarrowArgs &a);
hipError_t launch_farrow_wave(const TypeKey &tk, bool fused, const FarrowArgs &a, hipStream_t s, const char **kname, int num_cus);

thread_local LaunchEvents g_launch_events;
static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

namespace {
struct OccKey { const void *fn; int dev; bool operator==(const OccKey &o) const { return fn == o.fn && dev == o.dev; } };
struct OccKeyHash { size_t operator()(const OccKey &k) const { return std::hash<const void *>()(k.fn) ^ (static_cast<size_t>(k.dev) * 0x9e3779b97f4a7c15ull); } };
struct OccEntry { unsigned block = 0; size_t lds = ~static_cast<size_t>(0), attr = 0; int per_cu = 0; };
}  // namespace

hipError_t occupancy_cached(const void *kfn, unsigned block, size_t lds, int *per_cu)
{
    static std::mutex m;
    static std::unordered_map<OccKey, OccEntry, OccKeyHash> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(m);
    OccEntry &e = cache[OccKey{kfn, dev}];
    if (lds > 48 * 1024 && lds > e.attr) {            // the limit only ever grows
        hipError_t err = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (err != hipSuccess) return err;
        e.attr = lds;
    }
    if (e.block != block || e.lds != lds) {
        int n = 0;
        hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kfn, static_cast<int>(block), lds);
        if (err != hipSuccess) return err;
        e.block = block; e.lds = lds; e.per_cu = n;
    }
    *per_cu = e.per_cu;
    return hipSuccess;
}
int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

namespace {

// RAII: make the filter's device current for the duration of an API call, then restore.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

bool device_is_gfx950(int dev)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

TypeKey type_key(const mrhip_filter *f)
{
    return TypeKey{dtype_is_f64(f->tx), f->r_f64, f->nc == 2};
}

size_t r_size(const mrhip_filter *f) { return f->r_f64 ? 8 : 4; }
size_t x_elt(const mrhip_filter *f) { return dtype_size(f->tx); }
size_t y_elt(const mrhip_filter *f) { return dtype_size(f->ty); }

// upload taps (tap dtype on the host) as R-typed device array; f32 -> f64 widening is exact and is
// what Julia's promotion does on every multiply (Real*Real / Real*Complex methods).
// The vector sits between two runs of kTapPad zero elements: fir_stream_rt_kernel reads whole blocks of taps around the
// ends of a window (it discards what the out-of-window ones produce) without a clamp per tap.
constexpr size_t kDecimTabBytes = 4 * 64 * sizeof(float);
// decim_lane_kernel (kernels_decim_lane.hip) reads the taps of a FIRDecimator 1//4 x 128 Float32 taps as four doubled, age-ordered columns:
// they sit behind the tap vector's second pad.  NULL: not that shape.
const float *decim_tab(const mrhip_filter *f)
{
    if (f->kind != MRHIP_FIR_DECIMATOR || f->M != 4 || f->hLen != 128 || f->th != MRHIP_F32 || f->r_f64 || !f->d_taps_alloc) return nullptr;
    return reinterpret_cast<const float *>(static_cast<const unsigned char *>(f->d_taps_alloc) + 128 * sizeof(float) + 2 * static_cast<size_t>(mrhip::kTapPad) * sizeof(float));
}

int upload_taps(mrhip_filter *f, const std::vector<unsigned char> &src, void **dptr, void **alloc)
{
    const size_t n = src.size() / dtype_scalar_size(f->th);
    const size_t pad = static_cast<size_t>(mrhip::kTapPad) * r_size(f);
    const size_t bytes = std::max<size_t>(n * r_size(f), 16) + 2 * pad + kDecimTabBytes;   // (+ decim_lane_kernel's tap columns: decim_tab)
    MRHIP_CHECK_HIP(hipMalloc(alloc, bytes));
    MRHIP_CHECK_HIP(hipMemset(*alloc, 0, bytes));
    *dptr = static_cast<unsigned char *>(*alloc) + pad;
    if (f->r_f64 && f->th == MRHIP_F32) {
        std::vector<double> w(n);
        const float *s = reinterpret_cast<const float *>(src.data());
        for (size_t i = 0; i < n; ++i) w[i] = static_cast<double>(s[i]);
        MRHIP_CHECK_HIP(hipMemcpy(*dptr, w.data(), n * 8, hipMemcpyHostToDevice));
    } else {
        MRHIP_CHECK_HIP(hipMemcpy(*dptr, src.data(), src.size(), hipMemcpyHostToDevice));
    }
    return MRHIP_OK;
}

int alloc_common(mrhip_filter *f)
{
    const size_t hbytes = std::max<size_t>(static_cast<size_t>(f->nch) * f->H * x_elt(f), 16);
    // Every later launch goes to a non-blocking stream, which the null stream's memsets are not ordered with: zero
    // on the filter's own stream and wait for it here (construction may block; nothing else in the library does).
    MRHIP_CHECK_HIP(hipStreamCreateWithFlags(&f->own_stream, hipStreamNonBlocking));
    for (int i = 0; i < 3; ++i) {
        MRHIP_CHECK_HIP(hipMalloc(&f->d_hist[i], hbytes));
        MRHIP_CHECK_HIP(hipMemsetAsync(f->d_hist[i], 0, hbytes, f->own_stream));   // history = zeros(historyLen), Filters.jl:177
    }
    MRHIP_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&f->d_counters), mrhip::kCounterBytes));
    MRHIP_CHECK_HIP(hipMemsetAsync(f->d_counters, 0, mrhip::kCounterBytes, f->own_stream));
    MRHIP_CHECK_HIP(hipStreamSynchronize(f->own_stream));
    {
        hipDeviceProp_t prop;
        MRHIP_CHECK_HIP(hipGetDeviceProperties(&prop, f->device));
        f->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        const char *fg = std::getenv("MRHIP_FORCE_GENERIC");
        f->force_generic = fg && fg[0] == '1';
    }
    MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->sched_copied, hipEventDisableTiming));
    MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->xs_event, hipEventDisableTiming));
    return rec_alloc(f);              // the stream state on the device (stream_state.hip), constructor state
}

// Is `stream` being captured into a HIP graph?  (The legacy null stream cannot be; while another stream captures in
// global mode the query itself fails for it, which means "no".)
// (the streams the library has seen capturing are remembered: any_capture_active)
static std::mutex g_capture_mutex;
static std::vector<hipStream_t> g_capture_streams;

bool stream_is_capturing(hipStream_t stream)
{
    if (!stream) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (cs != hipStreamCaptureStatusActive) return false;
    std::lock_guard<std::mutex> g(g_capture_mutex);
    if (std::find(g_capture_streams.begin(), g_capture_streams.end(), stream) == g_capture_streams.end()) g_capture_streams.push_back(stream);
    return true;
}

// Is a stream the library has been called on still being captured?  A device-wide wait (what the host-side view of a filter whose calls
// were captured once needs: its replays run on streams the library never saw) would invalidate that capture -- in the relaxed capture
// mode too, as measured: rec_pull refuses instead.
bool any_capture_active_impl()
{
    std::lock_guard<std::mutex> g(g_capture_mutex);
    bool active = false;
    for (size_t i = 0; i < g_capture_streams.size();) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool ok = hipStreamIsCapturing(g_capture_streams[i], &cs) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        if (ok && cs == hipStreamCaptureStatusActive) { active = true; ++i; }
        else g_capture_streams.erase(g_capture_streams.begin() + static_cast<long>(i));    // (ended, or the stream is gone)
    }
    return active;
}

// Order this call after whatever the filter enqueued before on another stream (see mrhip_filter::last_stream).
int adopt_stream(mrhip_filter *f, hipStream_t stream)
{
    if (f->last_stream_valid && f->last_stream != stream) {
        if (hipEventRecord(f->xs_event, f->last_stream) == hipSuccess) {
            MRHIP_CHECK_HIP(hipStreamWaitEvent(stream, f->xs_event, 0));
        } else {                       // the earlier stream no longer exists: everything on the device is older
            (void)hipGetLastError();
            MRHIP_CHECK_HIP(device_sync_relaxed());
        }
    }
    f->last_stream = stream;
    f->last_stream_valid = true;
    return MRHIP_OK;
}

// Wait (on the host) for the filter's own enqueued work only -- not for the whole device.
int drain_filter(mrhip_filter *f)
{
    if (f->last_stream_valid && hipStreamSynchronize(f->last_stream) != hipSuccess) {
        (void)hipGetLastError();
        MRHIP_CHECK_HIP(device_sync_relaxed());
    }
    return MRHIP_OK;
}

int check_create_args(const void *h, int64_t hLen, int th, int tx, int64_t nch, int device, mrhip_filter **out)
{
    if (!out) return fail(MRHIP_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!h || hLen < 1) return fail(MRHIP_ERR_INVALID_ARG, "h must hold at least one tap");
    if (th != MRHIP_F32 && th != MRHIP_F64)
        return fail(MRHIP_ERR_UNSUPPORTED, "taps must be Float32 or Float64 (complex taps are not supported)");
    if (tx < MRHIP_F32 || tx > MRHIP_C128) return fail(MRHIP_ERR_INVALID_ARG, "bad sample dtype");
    if (nch < 1) return fail(MRHIP_ERR_INVALID_ARG, "nchannels must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MRHIP_ERR_NO_DEVICE, "no HIP device visible; libmultirate_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MRHIP_ERR_INVALID_ARG, "device ordinal out of range");
    if (!device_is_gfx950(device))
        return fail(MRHIP_ERR_NO_DEVICE, "device is not gfx950 (MI355X); kernels are built for gfx950 only");
    return MRHIP_OK;
}

// Descriptor staging of the multi-stream launches a filter leads (mrhip_filt_device_multi; period blocks below): pinned host +
// device memory, one event that guards the pinned half against the upload still in flight.
int multi_staging(mrhip_filter *f, size_t bytes)
{
    if (bytes > f->multi_cap) {
        if (f->multi_in_flight) { MRHIP_CHECK_HIP(hipEventSynchronize(f->multi_ev)); f->multi_in_flight = false; }
        if (int rc = drain_filter(f)) return rc;
        if (f->multi_pin) (void)hipHostFree(f->multi_pin);
        if (f->multi_dev) (void)hipFree(f->multi_dev);
        f->multi_pin = f->multi_dev = nullptr; f->multi_cap = 0;
        MRHIP_CHECK_HIP(hipHostMalloc(&f->multi_pin, bytes * 2, hipHostMallocDefault));
        MRHIP_CHECK_HIP(hipMalloc(&f->multi_dev, bytes * 2));
        f->multi_cap = bytes * 2;
        if (!f->multi_ev) MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->multi_ev, hipEventDisableTiming));
    }
    if (f->multi_in_flight) { MRHIP_CHECK_HIP(hipEventSynchronize(f->multi_ev)); f->multi_in_flight = false; }
    return MRHIP_OK;
}

// FIRRational / FIRInterpolator with L > 512 on the output-pair kernel: one stream descriptor per BLOCK of the period
// (kernels_rational_opair.hip: plan_rational_opair_blocks).  Block 0 carries the call's bookkeeping (record, shiftin!).
hipError_t launch_opair_blocks(mrhip_filter *f, const TypeKey &tk, bool fused, const PolyArgs &a, hipStream_t s, const char **kname, bool *launched)
{
    *launched = false;
    PairArgs pa;
    dim3 block;
    size_t lds = 0;
    int nblocks = 0;
    if (!plan_rational_opair_blocks(tk, fused, a, f->num_cus, &pa, &block, &lds, &nblocks)) return hipSuccess;
    const size_t bytes = static_cast<size_t>(nblocks) * sizeof(MultiDesc);
    if (multi_staging(f, bytes) != MRHIP_OK) return hipErrorOutOfMemory;
    MultiDesc *d = static_cast<MultiDesc *>(f->multi_pin);
    const size_t osz = dtype_scalar_size(f->ty) * static_cast<size_t>(f->nc);
    unsigned steps_max = 0;
    for (int b = 0; b < nblocks; ++b) {
        MultiDesc &m = d[b];
        const long long first = static_cast<long long>(b) * pa.P;                      // the block's first output within the period
        const long long Pb = std::min<long long>(pa.P, static_cast<long long>(pa.Sout) - first);
        const long long n_rel = a.n_out - first;                                         // outputs from the block's first one on
        const long long spc = n_rel > 0 ? (n_rel + pa.Sout - 1) / pa.Sout : 0;
        const long long u0b = a.u0 + first * a.M;
        m.x = a.x; m.y = static_cast<unsigned char *>(a.y) + static_cast<size_t>(first) * osz;
        m.hist = a.hist; m.hist_new = a.hist_new; m.taps = a.taps; m.rec = b == 0 ? a.rec : nullptr;
        m.x_stride = a.x_stride; m.y_stride = a.y_stride; m.x_len = a.x_len; m.n_out = std::max<long long>(n_rel, 0);
        m.u0 = u0b; m.d0 = a.d0; m.phi_end = a.phi_end; m.d_end = a.d_end;
        m.steps_per_channel = static_cast<unsigned>(std::max<long long>(spc, 1));
        m.total_steps = static_cast<unsigned>(spc * a.nch);
        m.spc_magic = spc <= 1 ? 0xffffffffu : static_cast<unsigned>((1ULL << 32) / static_cast<unsigned long long>(spc));
        m.nch = b == 0 ? a.nch : 0;                                                      // (shiftin! and the record: once per call)
        m.P_blk = static_cast<int>(Pb);
        m.q0 = static_cast<int>((u0b / a.L) & ~1LL);                                    // even: the lanes' run starts keep their parity
        steps_max = std::max(steps_max, m.total_steps);
    }
    hipError_t e = hipMemcpyAsync(f->multi_dev, f->multi_pin, bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    e = hipEventRecord(f->multi_ev, s);
    if (e != hipSuccess) return e;
    f->multi_in_flight = true;
    PolyArgs am = a;
    am.multi = static_cast<const MultiDesc *>(f->multi_dev); am.multi_n = nblocks;
    pa.total_steps = steps_max;
    e = launch_rational_opair(fused, am, pa, block, lds, s, kname, f->num_cus, f->d_counters);
    *launched = e == hipSuccess;
    return e;
}

}  // namespace
// kernels_interp_lane.hip: FIRInterpolator 4//1, 32 taps per phase, ComplexF32: a lane per channel (plans and launches; false: not its call)
bool try_launch_interp_lane(const TypeKey &tk, bool fused, const PolyArgs &a, unsigned *counters, hipStream_t s, const char **kname, int num_cus, hipError_t *err);
// (kernels_decim_lane.hip: FIRDecimator 1//4 x 128 taps, ComplexF32, long calls -- a lane per channel, the 32 outputs in flight in registers)
void decim_lane_table(const float *taps_oldest_first, float *tab);
bool try_launch_decim_lane(const TypeKey &tk, bool fused, const PolyArgs &a, const float *tab, unsigned *counters, hipStream_t s, const char **kname, int num_cus, hipError_t *err);
// (kernels_arb_window.hip: FIRArbitrary, Float64, 32 taps per phase, long calls -- a lane per channel, the window in registers)
bool try_launch_arb_window(const TypeKey &tk, bool fused, const ArbArgs &a, double rate, unsigned *counters, hipStream_t s, const char **kname, int num_cus, hipError_t *err);
namespace {

// Kernel selection for the rational family.  Tuned kernels are tried first; the universal
// one-thread-per-output kernel accepts everything.
hipError_t launch_poly(const mrhip_filter *f, const TypeKey &tk, bool fused, const PolyArgs &a, hipStream_t s,
                       const char **kname, bool *did_shiftin, unsigned *counters, bool *rec_written)
{
    *did_shiftin = false;
    *rec_written = false;             // the pair kernels and the universal kernel file the call's end state in the device record
    if (!f->force_generic) {
        if (a.L == 1) {
            hipError_t ed = hipSuccess;           // (shiftin! and the record are the caller's: did_shiftin / rec_written stay false)
            if (try_launch_decim_lane(tk, fused, a, decim_tab(f), counters, s, kname, f->num_cus, &ed)) return ed;
            PairArgs spa;
            dim3 sblock;
            size_t slds = 0;
            if (plan_fir_stream(tk, a, f->num_cus, &spa, &sblock, &slds)) {
                *rec_written = true;
                *did_shiftin = a.H > 0;          // its loader waves write the call-end history themselves
                return launch_fir_stream(fused, a, spa, sblock, slds, s, kname, f->num_cus, counters);
            }
        }
        {
            hipError_t el = hipSuccess;           // (shiftin! and the record are the caller's: did_shiftin / rec_written stay false)
            if (try_launch_interp_lane(tk, fused, a, counters, s, kname, f->num_cus, &el)) return el;
        }
        {
            PairArgs pa;
            dim3 block;
            size_t lds = 0;
            if (plan_rational_opair(tk, fused, a, f->num_cus, &pa, &block, &lds)) {
                *rec_written = true;
                *did_shiftin = a.H > 0;
                return launch_rational_opair(fused, a, pa, block, lds, s, kname, f->num_cus, counters);
            }
            bool launched = false;                 // L > 512: a workgroup per block of the period
            const hipError_t eb = launch_opair_blocks(const_cast<mrhip_filter *>(f), tk, fused, a, s, kname, &launched);
            if (eb != hipSuccess) return eb;
            if (launched) { *rec_written = true; *did_shiftin = a.H > 0; return hipSuccess; }
        }
        TileArgs ta;
        dim3 grid, block;
        size_t lds = 0;
        if (plan_phase_stationary(tk, a, f->num_cus, &ta, &grid, &block, &lds))
            return launch_poly_phase_stationary(tk, fused, a, ta, grid, block, lds, s, kname, f->num_cus);
        {   // any tapsPerPhi / any L: tap bank and sample tile in LDS
            ArbTileArgs tt;
            size_t tl = 0;
            if (plan_poly_tiled(tk, a, f->num_cus, &tt, &tl))
                return launch_poly_tiled(tk, fused, a, tt, tl, s, kname, f->num_cus);
        }
    }
    *rec_written = true;
    return launch_poly_generic(tk, fused, a, s, kname);
}

}  // namespace

bool any_capture_active() { return any_capture_active_impl(); }
}  // namespace mrhip

using namespace mrhip;

extern "C" {

int mrhip_abi_version(void) { return MRHIP_ABI_VERSION; }
const char *mrhip_last_error(void) { return g_last_error.c_str(); }

int mrhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) ok += device_is_gfx950(d) ? 1 : 0;
    return ok;
}

int64_t mrhip_taps2pfb(const void *h, int64_t hLen, int tap_dtype, int64_t Nphi, void *pfb)
{
    if (!h || hLen < 1 || Nphi < 1) return -1;
    return taps2pfb(h, hLen, tap_dtype, Nphi, pfb);
}
int64_t mrhip_nextphase(int64_t p, int64_t L, int64_t M) { return nextphase(p, L, M); }
int64_t mrhip_outputlength_ratio(int64_t n, int64_t L, int64_t M, int64_t phi) { return outputlength_ratio(n, L, M, phi); }
int64_t mrhip_inputlength_ratio(int64_t n, int64_t L, int64_t M, int64_t phi) { return inputlength_ratio(n, L, M, phi); }

int mrhip_output_dtype(int th, int tx)
{
    const bool f64 = dtype_is_f64(th) || dtype_is_f64(tx);
    if (dtype_is_complex(tx)) return f64 ? MRHIP_C128 : MRHIP_C64;
    return f64 ? MRHIP_F64 : MRHIP_F32;
}

int mrhip_create_rational(const void *h, int64_t hLen, int th, int64_t num, int64_t den, int tx, int64_t nch,
                          int device, mrhip_filter **out)
{
    if (int rc = check_create_args(h, hLen, th, tx, nch, device, out)) return rc;
    if (num < 1 || den < 1) return fail(MRHIP_ERR_INVALID_ARG, "ratio must be positive");
    const int64_t g = std::gcd(num, den);
    const int64_t L = num / g, M = den / g;
    if (L > 0x7fffffff || M > 0x7fffffff || hLen > 0x7fffffff)
        return fail(MRHIP_ERR_INVALID_ARG, "L, M and hLen must fit in 31 bits");

    DeviceGuard guard(device);
    if (!guard.ok) return fail(MRHIP_ERR_HIP, "hipSetDevice failed");
    auto *f = new mrhip_filter();
    f->th = th; f->tx = tx; f->ty = mrhip_output_dtype(th, tx);
    f->nc = dtype_is_complex(tx) ? 2 : 1;
    f->r_f64 = dtype_is_f64(f->ty);
    f->nch = nch; f->hLen = hLen; f->L = L; f->M = M; f->device = device;
    const size_t es = dtype_scalar_size(th);

    if (L == 1) {            // STANDARD (Filters.jl:163-165) / DECIMATOR (:166-168): h = flipud(h)
        f->kind = M == 1 ? MRHIP_FIR_STANDARD : MRHIP_FIR_DECIMATOR;
        f->Nphi = 1; f->T = hLen; f->H = hLen - 1;
        f->h_taps.resize(static_cast<size_t>(hLen) * es);
        taps2pfb(h, hLen, th, 1, f->h_taps.data());   // one column, reversed == flipud
    } else {                 // INTERPOLATOR (:169-171) / RATIONAL (:172-174)
        f->kind = M == 1 ? MRHIP_FIR_INTERPOLATOR : MRHIP_FIR_RATIONAL;
        f->Nphi = L;
        f->T = taps2pfb(h, hLen, th, L, nullptr);
        f->H = f->T - 1;
        f->h_taps.resize(static_cast<size_t>(f->T) * L * es);
        taps2pfb(h, hLen, th, L, f->h_taps.data());
    }
    int rc = upload_taps(f, f->h_taps, &f->d_taps, &f->d_taps_alloc);
    if (!rc && decim_tab(f)) {
        float tab[4 * 64];
        mrhip::decim_lane_table(reinterpret_cast<const float *>(f->h_taps.data()), tab);
        if (hipMemcpy(const_cast<float *>(decim_tab(f)), tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess) rc = fail(MRHIP_ERR_HIP, "hipMemcpy failed");
    }
    if (!rc) rc = alloc_common(f);
    if (rc) { mrhip_destroy(f); return rc; }
    *out = f;
    return MRHIP_OK;
}

int mrhip_create_arbitrary(const void *h, int64_t hLen, int th, double rate, int64_t Nphi, int tx, int64_t nch,
                           int device, mrhip_filter **out)
{
    if (int rc = check_create_args(h, hLen, th, tx, nch, device, out)) return rc;
    if (!(rate > 0.0)) return fail(MRHIP_ERR_INVALID_ARG, "rate must be greater than 0");
    if (Nphi < 1 || Nphi > 0x7fffffff || hLen > 0x7fffffff) return fail(MRHIP_ERR_INVALID_ARG, "bad Nphi");

    DeviceGuard guard(device);
    if (!guard.ok) return fail(MRHIP_ERR_HIP, "hipSetDevice failed");
    auto *f = new mrhip_filter();
    f->kind = MRHIP_FIR_ARBITRARY;
    f->th = th; f->tx = tx; f->ty = mrhip_output_dtype(th, tx);
    f->nc = dtype_is_complex(tx) ? 2 : 1;
    f->r_f64 = dtype_is_f64(f->ty);
    f->nch = nch; f->hLen = hLen; f->L = Nphi; f->M = 1; f->Nphi = Nphi; f->device = device;
    f->rate = rate;
    f->delta = static_cast<double>(Nphi) / rate;   // Δ = N𝜙/rate, Filters.jl:113
    const size_t es = dtype_scalar_size(th);

    // dh = [diff(h), 0] in the tap type (Filters.jl:106)
    std::vector<unsigned char> dh(static_cast<size_t>(hLen) * es, 0);
    if (th == MRHIP_F32) {
        const float *s = static_cast<const float *>(h);
        float *d = reinterpret_cast<float *>(dh.data());
        for (int64_t i = 0; i + 1 < hLen; ++i) d[i] = s[i + 1] - s[i];
    } else {
        const double *s = static_cast<const double *>(h);
        double *d = reinterpret_cast<double *>(dh.data());
        for (int64_t i = 0; i + 1 < hLen; ++i) d[i] = s[i + 1] - s[i];
    }
    f->T = taps2pfb(h, hLen, th, Nphi, nullptr);
    f->H = f->T - 1;
    f->h_taps.resize(static_cast<size_t>(f->T) * Nphi * es);
    f->h_dtaps.resize(static_cast<size_t>(f->T) * Nphi * es);
    taps2pfb(h, hLen, th, Nphi, f->h_taps.data());
    taps2pfb(dh.data(), hLen, th, Nphi, f->h_dtaps.data());

    int rc = upload_taps(f, f->h_taps, &f->d_taps, &f->d_taps_alloc);
    if (!rc) rc = upload_taps(f, f->h_dtaps, &f->d_dtaps, &f->d_dtaps_alloc);
    if (!rc) rc = alloc_common(f);
    if (rc) { mrhip_destroy(f); return rc; }
    sched_configure(f);
    *out = f;
    return MRHIP_OK;
}

static int create_farrow_common(const std::vector<double> &pnfb_in, int64_t hLen, int th, double rate, int64_t Nphi,
                                int64_t polyorder, int tx, int64_t nch, int device, mrhip_filter **out)
{
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MRHIP_ERR_HIP, "hipSetDevice failed");
    auto *f = new mrhip_filter();
    f->kind = MRHIP_FIR_FARROW;
    f->th = th; f->tx = tx; f->ty = mrhip_output_dtype(th, tx);
    f->nc = dtype_is_complex(tx) ? 2 : 1;
    f->r_f64 = dtype_is_f64(f->ty);
    f->nch = nch; f->hLen = hLen; f->L = Nphi; f->M = 1; f->Nphi = Nphi; f->device = device;
    f->rate = rate;
    f->delta = static_cast<double>(Nphi) / rate;   // Δ = N𝜙/rate, Filters.jl:143
    f->polyorder = polyorder;
    f->T = (hLen + Nphi - 1) / Nphi;
    f->H = f->T - 1;
    f->h_pnfb = pnfb_in;
    // Poly{T} storage: coefficients live in the tap type (Filters.jl:313 Array(Poly{T}, ...))
    if (th == MRHIP_F32) for (double &c : f->h_pnfb) c = static_cast<double>(static_cast<float>(c));
    int rc = MRHIP_OK;
    if (hipMalloc(reinterpret_cast<void **>(&f->d_pnfb), f->h_pnfb.size() * sizeof(double)) != hipSuccess ||
        hipMemcpy(f->d_pnfb, f->h_pnfb.data(), f->h_pnfb.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(MRHIP_ERR_HIP, "uploading the polynomial filter bank failed");
    if (!rc && f->T <= 32) {        // the same bank degree-major, padded to 32 taps with zeros (kernels_farrow_wave.hip)
        std::vector<double> t(static_cast<size_t>(polyorder + 1) * 32, 0.0);
        for (int64_t i = 0; i < f->T; ++i)
            for (int64_t j = 0; j <= polyorder; ++j) t[static_cast<size_t>(j) * 32 + i] = f->h_pnfb[static_cast<size_t>(i) * (polyorder + 1) + j];
        if (hipMalloc(reinterpret_cast<void **>(&f->d_pnfb_t), t.size() * sizeof(double)) != hipSuccess ||
            hipMemcpy(f->d_pnfb_t, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(MRHIP_ERR_HIP, "uploading the polynomial filter bank failed");
    }
    if (!rc) rc = alloc_common(f);
    if (rc) { mrhip_destroy(f); return rc; }
    sched_configure(f);
    *out = f;
    return MRHIP_OK;
}

int mrhip_create_farrow(const void *h, int64_t hLen, int th, double rate, int64_t Nphi, int64_t polyorder, int tx,
                        int64_t nch, int device, mrhip_filter **out)
{
    if (int rc = check_create_args(h, hLen, th, tx, nch, device, out)) return rc;
    if (!(rate > 0.0)) return fail(MRHIP_ERR_INVALID_ARG, "rate must be greater than 0");
    if (Nphi < 1 || Nphi > 0x7fffffff || hLen > 0x7fffffff) return fail(MRHIP_ERR_INVALID_ARG, "bad Nphi");
    if (polyorder < 0 || polyorder > 32 || polyorder + 1 > Nphi)
        return fail(MRHIP_ERR_INVALID_ARG, "polyorder must be in 0..min(32, Nphi-1)");
    // pfb = taps2pfb(h, Nphi); one polynomial per ROW (Filters.jl:139-140, :315-318)
    const size_t es = dtype_scalar_size(th);
    const int64_t T = taps2pfb(h, hLen, th, Nphi, nullptr);
    std::vector<unsigned char> pfb(static_cast<size_t>(T) * Nphi * es);
    taps2pfb(h, hLen, th, Nphi, pfb.data());
    std::vector<double> pn(static_cast<size_t>(T) * (polyorder + 1)), row(static_cast<size_t>(Nphi));
    for (int64_t i = 0; i < T; ++i) {
        for (int64_t c = 0; c < Nphi; ++c)          // element (row i, column c) of the column-major bank
            row[static_cast<size_t>(c)] = th == MRHIP_F32 ? static_cast<double>(reinterpret_cast<const float *>(pfb.data())[c * T + i])
                                                          : reinterpret_cast<const double *>(pfb.data())[c * T + i];
        if (!polyfit_rows(row.data(), Nphi, static_cast<int>(polyorder), &pn[static_cast<size_t>(i) * (polyorder + 1)]))
            return fail(MRHIP_ERR_INVALID_ARG, "polynomial fit is rank deficient");
    }
    return create_farrow_common(pn, hLen, th, rate, Nphi, polyorder, tx, nch, device, out);
}

int mrhip_create_farrow_pnfb(const double *pnfb, int64_t hLen, int th, double rate, int64_t Nphi, int64_t polyorder,
                             int tx, int64_t nch, int device, mrhip_filter **out)
{
    if (int rc = check_create_args(pnfb, hLen, th, tx, nch, device, out)) return rc;
    if (!(rate > 0.0)) return fail(MRHIP_ERR_INVALID_ARG, "rate must be greater than 0");
    if (Nphi < 1 || Nphi > 0x7fffffff || hLen > 0x7fffffff) return fail(MRHIP_ERR_INVALID_ARG, "bad Nphi");
    if (polyorder < 0 || polyorder > 32) return fail(MRHIP_ERR_INVALID_ARG, "polyorder must be in 0..32");
    const int64_t T = (hLen + Nphi - 1) / Nphi;
    std::vector<double> pn(pnfb, pnfb + static_cast<size_t>(T) * (polyorder + 1));
    return create_farrow_common(pn, hLen, th, rate, Nphi, polyorder, tx, nch, device, out);
}

int mrhip_polyfit(const double *y, int64_t n, int64_t polyorder, double *coef)
{
    if (!y || !coef || polyorder < 0 || polyorder > 64) return fail(MRHIP_ERR_INVALID_ARG, "bad polyfit arguments");
    if (!polyfit_rows(y, n, static_cast<int>(polyorder), coef)) return fail(MRHIP_ERR_INVALID_ARG, "polynomial fit is rank deficient");
    return MRHIP_OK;
}

int mrhip_get_pnfb(const mrhip_filter *f, double *host_out)
{
    if (!f || !host_out) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    if (f->kind != MRHIP_FIR_FARROW) return fail(MRHIP_ERR_INVALID_ARG, "not a FIRFarrow filter");
    std::memcpy(host_out, f->h_pnfb.data(), f->h_pnfb.size() * sizeof(double));
    return MRHIP_OK;
}

int mrhip_farrow_tapsforphase(const mrhip_filter *f, double phase, void *host_out)
{
    if (!f || !host_out) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    if (f->kind != MRHIP_FIR_FARROW) return fail(MRHIP_ERR_INVALID_ARG, "not a FIRFarrow filter");
    if (!(phase >= 0.0 && phase <= static_cast<double>(f->Nphi) + 1.0))
        return fail(MRHIP_ERR_INVALID_ARG, "phase must be >= 0 and <= Nphi+1");      // Filters.jl:765
    const int64_t P = f->polyorder;
    for (int64_t i = 0; i < f->T; ++i) {
        const double *c = &f->h_pnfb[static_cast<size_t>(i) * (P + 1)];
        double yv = c[P];
        for (int64_t j = P - 1; j >= 0; --j) { const double t = phase * yv; yv = c[j] + t; }
        if (f->th == MRHIP_F32) static_cast<float *>(host_out)[i] = static_cast<float>(yv);
        else static_cast<double *>(host_out)[i] = yv;
    }
    return MRHIP_OK;
}

void mrhip_destroy(mrhip_filter *f)
{
    if (!f) return;
    // a ring the filter still feeds: its resident kernel reads the taps and the history freed below (and hipFree would first wait for the
    // kernel's idle deadline), its close would write into the freed object.  Shut it down; the handle stays valid for mrhip_ring_close.
    if (f->ring) (void)ring_shutdown(f->ring);
    DeviceGuard guard(f->device);
    ring_cache_free(f);
    if (f->captured) (void)device_sync_relaxed();   // replays of a graph that holds this filter's calls ran on streams the library never saw
    (void)drain_filter(f);                       // this filter's work only; other streams of the process keep running
    for (hipStream_t st : {f->own_stream, f->s_in, f->s_out})
        if (st) (void)hipStreamSynchronize(st);
    for (void *p : {f->d_taps_alloc, f->d_dtaps_alloc, static_cast<void *>(f->d_pnfb), static_cast<void *>(f->d_pnfb_t), f->d_hist[0], f->d_hist[1], f->d_hist[2], static_cast<void *>(f->d_counters), f->d_sched_n, f->d_sched_acc,
                    f->d_xbuf[0], f->d_xbuf[1], f->d_ybuf[0], f->d_ybuf[1]})
        if (p) (void)hipFree(p);
    if (f->pin_n) (void)hipHostFree(f->pin_n);
    if (f->pin_acc) (void)hipHostFree(f->pin_acc);
    if (f->multi_pin) (void)hipHostFree(f->multi_pin);
    if (f->multi_dev) (void)hipFree(f->multi_dev);
    if (f->multi_ev) (void)hipEventDestroy(f->multi_ev);
    sched_free(f);
    rec_free(f);
    for (hipStream_t st : {f->own_stream, f->s_in, f->s_out})
        if (st) (void)hipStreamDestroy(st);
    if (f->ev_chain) (void)hipEventDestroy(f->ev_chain);
    if (f->ev_sdirty) (void)hipEventDestroy(f->ev_sdirty);
    for (hipEvent_t e : {f->sched_copied, f->xs_event, f->ev_in[0], f->ev_in[1], f->ev_k[0], f->ev_k[1], f->ev_out[0], f->ev_out[1]})
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : f->ev_pool) (void)hipEventDestroy(e);
    delete f;
}

// The host's state written into the device record, in stream order behind whatever the filter enqueued last (the stream of
// its last call; when that stream no longer exists -- torch side streams come and go -- behind the whole device, on the
// filter's own stream).
// the filter's schedule stream behind a chained call's schedule, which ran on the caller's stream (filt_device_one)
// The schedule stream behind what ran on a caller's stream (chain_pending / chain_stream): the event is recorded NOW -- it then covers at least
// what it has to -- and not by every small call (one host call each: profiles/r05/experiments.md O).  A stream that no longer exists has
// nothing pending the device could still be running for long: the device is waited for instead.
static int chain_event_now(mrhip_filter *f)
{
    if (!f->ev_chain) MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->ev_chain, hipEventDisableTiming));
    if (hipEventRecord(f->ev_chain, f->chain_stream) == hipSuccess) {
        MRHIP_CHECK_HIP(hipStreamWaitEvent(f->s_sched, f->ev_chain, 0));
    } else {
        (void)hipGetLastError();
        MRHIP_CHECK_HIP(device_sync_relaxed());
    }
    f->chain_pending = false;
    return MRHIP_OK;
}

static int sched_stream_behind_chain(mrhip_filter *f)
{
    if (f->s_sched && f->chain_pending) {
        if (int rc = chain_event_now(f)) return rc;
    }
    return MRHIP_OK;
}

static int push_state(mrhip_filter *f)
{
    if (f->s_sched) {
        if (int rc = sched_stream_behind_chain(f)) return rc;
        f->async_pending = true;
        f->sched_dirty = true;
        return rec_push(f, f->s_sched);
    }   // FIRArbitrary / FIRFarrow: every write of the record, in program order
    hipStream_t s = f->last_stream_valid ? f->last_stream : f->own_stream;
    if (rec_push(f, s) != MRHIP_OK) {
        (void)hipGetLastError();
        MRHIP_CHECK_HIP(device_sync_relaxed());
        s = f->own_stream;
        if (int rc = rec_push(f, s)) return rc;
    }
    f->last_stream = s; f->last_stream_valid = true;
    return MRHIP_OK;
}

// the host's copy of the stream state is stale after calls nobody collected (asynchronous calls, graph replays)
static int fresh(const mrhip_filter *f)
{
    if (f->mirror_valid) return MRHIP_OK;
    DeviceGuard guard(f->device);
    return rec_pull(const_cast<mrhip_filter *>(f));
}

int64_t mrhip_outputlength(const mrhip_filter *f, int64_t n)
{
    if (!f) return -1;
    if (f->kind != MRHIP_FIR_STANDARD && f->kind != MRHIP_FIR_INTERPOLATOR && fresh(f)) return -1;
    switch (f->kind) {
    case MRHIP_FIR_STANDARD: return n;                                                   // Filters.jl:359
    case MRHIP_FIR_INTERPOLATOR: return f->L * n;                                        // :363
    case MRHIP_FIR_DECIMATOR: return outputlength_ratio(n - f->inputDeficit + 1, 1, f->M, 1);   // :367
    case MRHIP_FIR_RATIONAL: return outputlength_ratio(n - f->inputDeficit + 1, f->L, f->M, f->phiIdx); // :371
    case MRHIP_FIR_ARBITRARY: return static_cast<int64_t>(std::ceil(static_cast<double>(n - f->inputDeficit + 1) * f->rate)); // :375
    case MRHIP_FIR_FARROW: return static_cast<int64_t>(std::ceil(static_cast<double>(n - f->inputDeficit + 1) * f->rate));    // :379
    }
    return -1;
}

// FIRArbitrary: evaluate (or reuse) the phase schedule for a call of x_len samples from the current state
static int64_t arb_schedule(mrhip_filter *f, int64_t x_len, ArbState *end_state)
{
    if (!(f->sched_cached && f->sched_xlen == x_len && f->sched_acc0 == f->phiAcc && f->sched_deficit0 == f->inputDeficit)) {
        if (f->sched_in_flight) { (void)hipEventSynchronize(f->sched_copied); f->sched_in_flight = false; }
        ArbState st{f->phiAcc, f->phiIdx, f->alpha, f->xIdx, f->inputDeficit};
        f->sched_acc0 = f->phiAcc; f->sched_deficit0 = f->inputDeficit; f->sched_xlen = x_len;
        f->sched_count = run_arbitrary_schedule(st, f->delta, f->Nphi, x_len, &f->sched_n, &f->sched_acc, f->mod_form);
        f->sched_end = st;
        f->sched_cached = true;
    }
    if (end_state) *end_state = f->sched_end;
    return f->sched_count;
}

int64_t mrhip_next_output_count(const mrhip_filter *f, int64_t n)
{
    if (!f || n < 0) return -1;
    if (fresh(f)) return -1;
    if (f->kind == MRHIP_FIR_ARBITRARY || f->kind == MRHIP_FIR_FARROW)   // the schedule is cached for the filt call that normally follows
        return arb_schedule(const_cast<mrhip_filter *>(f), n, nullptr);
    return plan_rational(f->kind, f->L, f->M, f->phiIdx, f->inputDeficit, n).n_out;
}

int64_t mrhip_advance_state(mrhip_filter *f, int64_t n)
{
    if (!f || n < 0) { (void)fail(MRHIP_ERR_INVALID_ARG, "advance_state: NULL filter or negative length"); return -1; }
    if (fresh(f)) return -1;
    DeviceGuard guard(f->device);
    hipStream_t s = f->last_stream_valid ? f->last_stream : f->own_stream;
    if (stream_is_capturing(s)) { (void)fail(MRHIP_ERR_UNSUPPORTED, "advance_state while the filter's stream is being captured"); return -1; }
    int64_t total;
    if (f->kind == MRHIP_FIR_ARBITRARY || f->kind == MRHIP_FIR_FARROW) {
        // The recurrence of update() (Filters.jl:663-673) over n samples, count and end state only.  Long stretches go through
        // the device-evaluated schedule (the same kernels a filt! call runs, without the filter kernel), launch-sized piece
        // by piece: 1e8 outputs in a few ms instead of 0.13 s of the host's serial loop (time sharding enters a stream at
        // its first sample this way, sharding.py); short ones, and rates the device evaluation does not cover, keep the loop.
        total = 0;
        int64_t left = n;
        const int64_t step = std::max<int64_t>(4096, static_cast<int64_t>(static_cast<double>(1LL << 24) / f->rate));
        while (left > 0) {
            const int64_t len = std::min(left, step);
            const int64_t est = len >= f->inputDeficit ? static_cast<int64_t>(std::ceil(static_cast<double>(len - f->inputDeficit + 1) * f->rate)) + 2 : 0;
            ArbState st{f->phiAcc, f->phiIdx, f->alpha, f->xIdx, f->inputDeficit};
            int64_t got = 0;
            if (sched_wants_device(f, est) && f->s_sched) {
                SchedOut so{};
                so.buf = f->flip; f->flip ^= 1;
                if (f->ev_filt_valid[so.buf] && hipStreamWaitEvent(f->s_sched, f->ev_filt[so.buf], 0) != hipSuccess) return -1;
                f->sched_dirty = true;
                if (sched_enqueue(f, len, est, INT64_MAX, nullptr, true, f->s_sched, &so)) return -1;
                for (;;) {
                    bool relaunch = false;
                    if (sched_collect(f, len, est, INT64_MAX, nullptr, f->s_sched, &so, &relaunch)) return -1;
                    if (!relaunch) break;
                }
                got = so.count; st = so.end;
                f->sched_drift = so.drift; f->sched_ksteps = so.ksteps;
                if (so.periodic || f->per_valid) f->per_pos = so.per_pos_end;
                f->memo_valid = false;                      // (the entries in the buffer belong to a call that was never made)
            } else {
                got = run_arbitrary_schedule(st, f->delta, f->Nphi, len, nullptr, nullptr, f->mod_form);    // (count only: no entries kept)
                sched_forget(f);
            }
            f->phiAcc = st.acc; f->phiIdx = st.phiIdx; f->alpha = st.alpha; f->xIdx = st.xIdx;   // Filters.jl:731-735
            f->inputDeficit = st.inputDeficit;
            total += got;
            left -= len;
        }
        f->sched_cached = false;
    } else {
        const CallPlan p = plan_rational(f->kind, f->L, f->M, f->phiIdx, f->inputDeficit, n);
        f->phiIdx = p.phi_end;                                              // Filters.jl:515-516, 571-572, 647-648
        f->inputDeficit = p.d_end;
        total = p.n_out;
    }
    if (push_state(f)) return -1;                                           // the device record follows, in stream order
    return total;
}

int64_t mrhip_inputlength(const mrhip_filter *f, int64_t n)
{
    if (!f) return -1;
    if (fresh(f)) return -1;
    switch (f->kind) {
    case MRHIP_FIR_STANDARD: return n;                                                    // Filters.jl:403
    case MRHIP_FIR_INTERPOLATOR: return inputlength_ratio(n, f->L, 1, 1);                 // :407
    // :412-416 reads a non-existent field; restated with the Rational method's intent (:418-422)
    case MRHIP_FIR_DECIMATOR: return inputlength_ratio(n, 1, f->M, 1) + f->inputDeficit - 1;
    case MRHIP_FIR_RATIONAL: return inputlength_ratio(n, f->L, f->M, f->phiIdx) + f->inputDeficit - 1;
    default: return -1;
    }
}

int mrhip_get_state(const mrhip_filter *f, mrhip_state *st)
{
    if (!f || !st) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    if (int rc = fresh(f)) return rc;
    st->kind = f->kind; st->tap_dtype = f->th; st->sample_dtype = f->tx; st->output_dtype = f->ty;
    st->nchannels = f->nch; st->hLen = f->hLen; st->interpolation = f->L; st->decimation = f->M;
    st->Nphi = f->Nphi; st->tapsPerPhi = f->T; st->historyLen = f->H;
    st->phiIdx = f->phiIdx; st->inputDeficit = f->inputDeficit; st->xIdx = f->xIdx;
    st->rate = f->rate; st->phiAccumulator = f->phiAcc; st->alpha = f->alpha; st->delta = f->delta;
    return MRHIP_OK;
}

int mrhip_set_state(mrhip_filter *f, int64_t phiIdx, int64_t inputDeficit, double phiAccumulator)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    if (f->ring_open) return fail(MRHIP_ERR_INVALID_ARG, "the filter feeds a ring (mrhip_ring_open): close it first");
    if (inputDeficit < 1) return fail(MRHIP_ERR_INVALID_ARG, "inputDeficit must be >= 1");
    DeviceGuard guard(f->device);
    hipStream_t s = f->last_stream_valid ? f->last_stream : f->own_stream;
    if (stream_is_capturing(s)) return fail(MRHIP_ERR_UNSUPPORTED, "set_state while the filter's stream is being captured");
    if (f->kind == MRHIP_FIR_ARBITRARY || f->kind == MRHIP_FIR_FARROW) {
        if (!(phiAccumulator >= 1.0) || !(phiAccumulator < static_cast<double>(f->Nphi) + 1.0))
            return fail(MRHIP_ERR_INVALID_ARG, "phiAccumulator must be in [1, Nphi+1)");
        f->phiAcc = phiAccumulator;
        f->phiIdx = static_cast<int64_t>(std::floor(phiAccumulator));
        f->alpha = phiAccumulator - static_cast<double>(f->phiIdx);
    } else {
        if (phiIdx < 1 || phiIdx > f->Nphi) return fail(MRHIP_ERR_INVALID_ARG, "phiIdx must be in 1..Nphi");
        f->phiIdx = phiIdx;
    }
    f->inputDeficit = inputDeficit;
    f->sched_cached = false;
    sched_forget(f);
    // the device record follows in stream order (behind the calls already enqueued, like the reference's assignment
    // behind the calls already made); whatever asynchronous calls left in the host's copy is overwritten with it
    f->mirror_valid = true;
    return push_state(f);
}

int mrhip_get_history(mrhip_filter *f, void *host_out)
{
    if (!f || !host_out) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    DeviceGuard guard(f->device);
    if (int rc = drain_filter(f)) return rc;     // the history is written by the filter's own last launch
    const size_t bytes = static_cast<size_t>(f->nch) * f->H * x_elt(f);
    if (bytes) MRHIP_CHECK_HIP(hipMemcpy(host_out, f->d_hist[f->hist_cur], bytes, hipMemcpyDeviceToHost));
    return MRHIP_OK;
}

int mrhip_set_history(mrhip_filter *f, const void *host_in)
{
    if (!f || !host_in) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    DeviceGuard guard(f->device);
    // stream-ordered behind the filter's earlier launches and ahead of its next one (which either runs on the same
    // stream or waits for it, adopt_stream); the host buffer is the caller's, so wait for the copy itself
    hipStream_t s = f->last_stream_valid ? f->last_stream : f->own_stream;
    if (stream_is_capturing(s)) return fail(MRHIP_ERR_UNSUPPORTED, "set_history while the filter's stream is being captured");
    const size_t bytes = static_cast<size_t>(f->nch) * f->H * x_elt(f);
    hist_leave_zeros(f);
    if (bytes && hipMemcpyAsync(f->d_hist[f->hist_cur], host_in, bytes, hipMemcpyHostToDevice, s) != hipSuccess) {
        // the stream of the filter's last call no longer exists (the caller destroyed it): everything it carried has
        // either run or gone with it -- order behind the whole device once and carry on on the filter's own stream
        (void)hipGetLastError();
        MRHIP_CHECK_HIP(device_sync_relaxed());
        s = f->own_stream;
        MRHIP_CHECK_HIP(hipMemcpyAsync(f->d_hist[f->hist_cur], host_in, bytes, hipMemcpyHostToDevice, s));
    }
    if (hipStreamSynchronize(s) != hipSuccess) {
        (void)hipGetLastError();
        MRHIP_CHECK_HIP(device_sync_relaxed());
        s = f->own_stream;
    }
    f->last_stream = s; f->last_stream_valid = true;
    return MRHIP_OK;
}

int mrhip_set_history_device(mrhip_filter *f, const void *dev_in, void *stream_)
{
    if (!f || !dev_in) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    DeviceGuard guard(f->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (stream_is_capturing(stream)) return fail(MRHIP_ERR_UNSUPPORTED, "set_history while the stream is being captured");
    if (int rc = adopt_stream(f, stream)) return rc;          // behind the filter's earlier launches, ahead of its next one
    const size_t bytes = static_cast<size_t>(f->nch) * f->H * x_elt(f);
    hist_leave_zeros(f);
    if (bytes) MRHIP_CHECK_HIP(hipMemcpyAsync(f->d_hist[f->hist_cur], dev_in, bytes, hipMemcpyDeviceToDevice, stream));
    return MRHIP_OK;
}

int mrhip_reset(mrhip_filter *f)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    if (f->ring_open) return fail(MRHIP_ERR_INVALID_ARG, "the filter feeds a ring (mrhip_ring_open): close it first");
    DeviceGuard guard(f->device);
    // No host wait: the zeroing is enqueued on the stream the filter last ran on, i.e. after its earlier launches;
    // the next filt call is on that stream too, or waits for it (adopt_stream).
    hipStream_t s = f->last_stream_valid ? f->last_stream : f->own_stream;
    const size_t bytes = static_cast<size_t>(f->nch) * f->H * x_elt(f);
    // history = zeros(historyLen) (Filters.jl:177): the read-only slot of zeros becomes the current one -- no memset, no launch -- unless a
    // captured graph has a slot baked in (its replays read THAT slot: zero it).  The hand-out counters are zero between launches by
    // their own protocol; they are zeroed once more by the kernel that pushes the record below (no launch of their own) when that
    // kernel runs on the filter's stream.
    const bool fast = !f->captured && MRHIP_ENV_INT("MRHIP_RESET_FAST", 1) != 0;
    const bool rec_on_s = !f->s_sched;                 // (FIRArbitrary / FIRFarrow push the record on their schedule stream)
    if (fast) f->hist_cur = 2;
    auto zero_on = [&](hipStream_t st) -> hipError_t {
        if (bytes && !fast) { hipError_t e = hipMemsetAsync(f->d_hist[f->hist_cur], 0, bytes, st); if (e != hipSuccess) return e; }
        if (fast && rec_on_s) return hipSuccess;
        return hipMemsetAsync(f->d_counters, 0, mrhip::kCounterBytes, st);
    };
    if (stream_is_capturing(s)) return fail(MRHIP_ERR_UNSUPPORTED, "reset while the filter's stream is being captured");
    if (zero_on(s) != hipSuccess) {
        // the stream of the filter's last call no longer exists (torch side streams come and go): like adopt_stream,
        // order behind the whole device once and carry on on the filter's own stream
        (void)hipGetLastError();
        MRHIP_CHECK_HIP(device_sync_relaxed());
        s = f->own_stream;
        MRHIP_CHECK_HIP(zero_on(s));
    }
    f->last_stream = s; f->last_stream_valid = true;
    f->phiIdx = 1; f->inputDeficit = 1; f->xIdx = 1; f->phiAcc = 1.0; f->alpha = 0.0;
    f->sched_cached = false;
    f->mirror_valid = true;
    // The stream restarts from the constructor state, i.e. on the very trajectory the device schedule's drift-per-step
    // estimate was measured on: keep it (no serial host prefix, full-size pieces at once; every piece is verified
    // anyway).  A stream that was found to cycle re-enters its cycle where the constructor state sits on it -- when it does
    // (rate 1.0: the cycle IS that state); otherwise forget the cycle (the prefix finds it again).
    if (f->per_valid && f->per_reset_pos >= 0 && MRHIP_ENV_INT("MRHIP_SCHED_KEEP_DRIFT", 1) != 0) f->per_pos = f->per_reset_pos;
    else if (f->per_valid || MRHIP_ENV_INT("MRHIP_SCHED_KEEP_DRIFT", 1) == 0) sched_forget(f);
    if (f->s_sched) {
        f->async_pending = true;
        if (int rc = sched_stream_behind_chain(f)) return rc;
    }
    if (f->s_sched) f->sched_dirty = true;
    if (f->s_sched) return rec_push(f, f->s_sched);                   // the device record: constructor state, in stream order
    if (fast) {
        // the record's kernel zeroes the counters too; a stream that no longer exists (torch side streams come and go): behind the
        // whole device once, on the filter's own stream
        if (rec_push(f, s, -1, -1, f->d_counters) != MRHIP_OK) {
            (void)hipGetLastError();
            MRHIP_CHECK_HIP(device_sync_relaxed());
            f->last_stream = f->own_stream;
            return rec_push(f, f->own_stream, -1, -1, f->d_counters);
        }
        return MRHIP_OK;
    }
    return rec_push(f, s);
}

int mrhip_set_numerics(mrhip_filter *f, int numerics)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    if (numerics != MRHIP_NUMERICS_STRICT && numerics != MRHIP_NUMERICS_FUSED)
        return fail(MRHIP_ERR_INVALID_ARG, "unknown numerics mode");
    f->numerics = numerics;
    return MRHIP_OK;
}

int mrhip_set_mod_form(mrhip_filter *f, int form)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    if (form != 0 && form != 1) return fail(MRHIP_ERR_INVALID_ARG, "mod form must be 0 (exact remainder) or 1 (rem(y + rem(x, y), y))");
    if (f->kind != MRHIP_FIR_ARBITRARY && f->kind != MRHIP_FIR_FARROW) return fail(MRHIP_ERR_INVALID_ARG, "not a FIRArbitrary / FIRFarrow filter");
    if (f->mod_form != form) {
        f->mod_form = form;
        f->sched_cached = false;
        sched_forget(f);           // drift estimate, cycle and memo belong to the other recurrence
        f->memo_valid = false;
    }
    return MRHIP_OK;
}

int mrhip_get_taps(mrhip_filter *f, int which, void *host_out)
{
    if (!f || !host_out) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    const auto &src = which ? f->h_dtaps : f->h_taps;
    if (src.empty()) return fail(MRHIP_ERR_INVALID_ARG, "filter has no such tap bank");
    std::memcpy(host_out, src.data(), src.size());
    return MRHIP_OK;
}

// ---------------------------------------------------------------------------------------
// the hot path
// ---------------------------------------------------------------------------------------
// Timing log (no-op unless timing is enabled).  Called in pairs around every compute launch; with a stride only
// every n-th launch is timed (timing costs stream time itself).  Two mechanisms:
//  * default: hipEventRecord before and after the launch on the launch stream (two marker packets, ~6 us of stream
//    time per timed launch; the interval reads ~5 us longer than the kernel's own begin/end timestamps in a
//    rocprofv3 kernel trace of the same launch);
//  * MRHIP_TIMING_ATTACH=1: the pair is attached to the kernel's own dispatch (launch_kernel() -> hipExtLaunchKernel).
//    Measured on MI355X / ROCm 7.2: same interval as the bracket, but the runtime serialises around such a launch
//    (8.8 us idle before, 4.8 us after in the kernel trace), so it disturbs the throughput more.  Kept for experiments.
static int timing_mark(mrhip_filter *f, hipStream_t stream)
{
    if (!f->timing) return MRHIP_OK;
    static const bool attach = [] { const char *v = std::getenv("MRHIP_TIMING_ATTACH"); return v && v[0] == '1'; }();
    if (f->timing_group > 1) {
        // group mode: ONE event pair around every `timing_group` consecutive compute launches (the gaps between the
        // launches of a group are inside the bracket -- a streaming cost -- and the marker packets' own stream time is
        // shared by the whole group); an incomplete last group is dropped by mrhip_timing_read
        if (!f->timing_open) {
            f->timing_open = true;
            if (f->timing_launch % f->timing_group == 0) {
                while (f->ev_pool.size() < f->ev_used + 2) {
                    hipEvent_t e = nullptr;
                    MRHIP_CHECK_HIP(hipEventCreate(&e));
                    f->ev_pool.push_back(e);
                }
                MRHIP_CHECK_HIP(hipEventRecord(f->ev_pool[f->ev_used], stream));
                f->ev_used += 1;                               // the stop event follows when the group is complete
            }
        } else {
            f->timing_open = false;
            if (++f->timing_launch % f->timing_group == 0) {
                MRHIP_CHECK_HIP(hipEventRecord(f->ev_pool[f->ev_used], stream));
                f->ev_used += 1;
            }
        }
        return MRHIP_OK;
    }
    if (!f->timing_open) {
        f->timing_open = true;
        const bool take = (f->timing_launch++ % f->timing_stride) == 0;
        f->ev_skip = !take;
        if (!take) return MRHIP_OK;
        while (f->ev_pool.size() < f->ev_used + 2) {
            hipEvent_t e = nullptr;
            MRHIP_CHECK_HIP(hipEventCreate(&e));
            f->ev_pool.push_back(e);
        }
        if (attach) {
            g_launch_events.start = f->ev_pool[f->ev_used];
            g_launch_events.stop = f->ev_pool[f->ev_used + 1];
        } else {
            MRHIP_CHECK_HIP(hipEventRecord(f->ev_pool[f->ev_used], stream));
        }
        f->ev_used += 2;
    } else {
        f->timing_open = false;
        if (f->ev_skip) return MRHIP_OK;
        if (attach) {
            if (g_launch_events.start) {          // no kernel consumed the pair (nothing to launch)
                g_launch_events = LaunchEvents{};
                f->ev_used -= 2;
            }
        } else {
            MRHIP_CHECK_HIP(hipEventRecord(f->ev_pool[f->ev_used - 1], stream));
        }
    }
    return MRHIP_OK;
}

static int ensure_sched_capacity(mrhip_filter *f, size_t n)
{
    if (n > f->pin_cap) {
        const size_t cap = std::max<size_t>(n + n / 4, 4096);
        if (f->sched_in_flight) { MRHIP_CHECK_HIP(hipEventSynchronize(f->sched_copied)); f->sched_in_flight = false; }
        if (f->pin_n) (void)hipHostFree(f->pin_n);
        if (f->pin_acc) (void)hipHostFree(f->pin_acc);
        f->pin_n = f->pin_acc = nullptr;
        MRHIP_CHECK_HIP(hipHostMalloc(&f->pin_n, cap * sizeof(int32_t), hipHostMallocDefault));
        MRHIP_CHECK_HIP(hipHostMalloc(&f->pin_acc, cap * sizeof(double), hipHostMallocDefault));
        f->pin_cap = cap;
    }
    if (n > f->d_sched_cap) {
        const size_t cap = std::max<size_t>(n + n / 4, 4096);
        if (int rc = drain_filter(f)) return rc;      // only this filter's kernels read the schedule buffers
        if (f->d_sched_n) (void)hipFree(f->d_sched_n);
        if (f->d_sched_acc) (void)hipFree(f->d_sched_acc);
        f->d_sched_n = f->d_sched_acc = nullptr;
        MRHIP_CHECK_HIP(hipMalloc(&f->d_sched_n, cap * sizeof(int32_t)));
        MRHIP_CHECK_HIP(hipMalloc(&f->d_sched_acc, cap * sizeof(double)));
        f->d_sched_cap = cap;
    }
    return MRHIP_OK;
}

// a-priori bound of n[last] - n[first] over `m` consecutive outputs of FIRArbitrary / FIRFarrow: every update() advances
// xIdx by floor((phase + delta) / N) (Filters.jl:666-668), so m - 1 steps advance it by at most (m - 1) / rate + 1; one more
// for the roundings of the Float64 recurrence.  Sizes the LDS sample tiles of a call whose schedule is still on its way.
static void span_bounds(double rate, int *spans)
{
    for (int z = 0; z < kSchedSpanSizes; ++z) {
        const double m = static_cast<double>((static_cast<long long>(kSchedSpanBase) << z) - 1);
        const double b = std::floor(m / rate) + 2.0;
        spans[z] = b < 2.0e9 ? static_cast<int>(b) : 0x7fffffff;
    }
}

// the largest output count a call of x_len samples can have, whatever the stream state (what y must hold for a call
// that is planned on the device)
static int64_t output_bound(const mrhip_filter *f, int64_t x_len)
{
    switch (f->kind) {
    case MRHIP_FIR_STANDARD: return x_len;
    case MRHIP_FIR_INTERPOLATOR: return f->L * x_len;
    case MRHIP_FIR_DECIMATOR: return outputlength_ratio(x_len, 1, f->M, 1);
    case MRHIP_FIR_RATIONAL: return outputlength_ratio(x_len, f->L, f->M, 1);
    default: return x_len > 0 ? static_cast<int64_t>(std::ceil(static_cast<double>(x_len) * f->rate)) + 2 : 0;   // Filters.jl:375-381 with inputDeficit = 1, + 2
    }
}

// Kernel selection for a DEVICE-PLANNED call of the rational family: the kernels that read the call record (the two pair
// kernels and the universal one); `a` holds the upper bounds the launch is sized with.  *P receives the outputs per step
// the plan kernel needs for the pair kernels' step walk.
static hipError_t launch_poly_dyn(mrhip_filter *f, const TypeKey &tk, bool fused, const PolyArgs &a, int64_t x_len, long long y_capacity,
                                  long long *count_dev, hipStream_t s, const char **kname, bool *did_shiftin, const DevCall *x_from = nullptr)
{
    *did_shiftin = false;
    if (!f->force_generic) {
        PairArgs pa;
        dim3 block;
        size_t lds = 0;
        if (a.L == 1 && plan_fir_stream(tk, a, f->num_cus, &pa, &block, &lds)) {
            hipError_t e = launch_poly_plan(f, x_len, pa.P, y_capacity, count_dev, s, x_from);
            if (e != hipSuccess) return e;
            *did_shiftin = a.H > 0;
            return launch_fir_stream(fused, a, pa, block, lds, s, kname, f->num_cus, f->d_counters);
        }
        if (a.L > 1 && plan_rational_opair(tk, fused, a, f->num_cus, &pa, &block, &lds)) {
            hipError_t e = launch_poly_plan(f, x_len, pa.P, y_capacity, count_dev, s, x_from);
            if (e != hipSuccess) return e;
            *did_shiftin = a.H > 0;
            return launch_rational_opair(fused, a, pa, block, lds, s, kname, f->num_cus, f->d_counters);
        }
    }
    // (a chained call on the universal kernel: its history update is a launch of its own, which takes the length from the call record too)
    hipError_t e = launch_poly_plan(f, x_len, 1, y_capacity, count_dev, s, x_from);
    if (e != hipSuccess) return e;
    return launch_poly_generic(tk, fused, a, s, kname);
}

// One launch-sized piece of a filt! call.  `continuation`: the piece continues a call whose earlier samples were already
// filtered (mrhip_filt_device splits calls longer than a launch can index; mrhip_filt_host cuts a call into staging
// pieces): the Vector seam's start-from-zero (support.jl:46) then applies to none of its outputs -- in the ONE reference
// call they all lie past the first hLen samples -- so that a split call equals the unsplit one down to the sign of a zero.
// `async`: nobody waits for this call (mrhip_filt_device_async): it is planned on the device from the device-resident
// stream state, like every call that is being captured into a HIP graph.
static int filt_device_one(mrhip_filter *f, const void *x, int64_t x_len, int64_t x_stride, void *y,
                           int64_t y_capacity, int64_t y_stride, int64_t *n_written, void *stream_, bool continuation,
                           bool async = false, long long *count_dev = nullptr, const DevCall *x_from = nullptr)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    if (n_written) *n_written = 0;
    if (x_len < 0 || y_capacity < 0) return fail(MRHIP_ERR_INVALID_ARG, "negative length");
    if (x_len > 0 && !x) return fail(MRHIP_ERR_INVALID_ARG, "x is NULL");
    if (x_len >= 0x7fffffffLL) return fail(MRHIP_ERR_INVALID_ARG, "a launch takes fewer than 2^31-1 samples per channel (internal)");
    if (f->nch > 1 && (x_stride < x_len)) return fail(MRHIP_ERR_INVALID_ARG, "x_stride < x_len");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceGuard guard(f->device);
    if (!guard.ok) return fail(MRHIP_ERR_HIP, "hipSetDevice failed");
    const TypeKey tk = type_key(f);
    const bool fused = f->numerics == MRHIP_NUMERICS_FUSED;
    const bool arb = f->kind == MRHIP_FIR_ARBITRARY || f->kind == MRHIP_FIR_FARROW;

    // Stream capture (HIP graphs).  A captured call is replayed with the arguments baked in at capture time while the
    // host object is NOT advanced by a replay: so it is planned ON THE DEVICE -- a one-lane plan kernel in front of the
    // filter kernel reads the stream state from the device record, leaves count and call-start state in the call record
    // the filter kernel reads, and advances the record -- and every replay continues the stream like the next call of a
    // plain loop, at ANY fixed chunk size and for every kind.  The history is written back into the slot it was read
    // from (the slot is baked in too).  The host fields move on as a shadow of ONE replay (so that the calls of a capture
    // see each other) and are re-read from the device before they are next used.
    const bool capturing = stream_is_capturing(stream);
    const bool dev_planned = async || capturing;
    if (capturing && f->async_pending)
        return fail(MRHIP_ERR_UNSUPPORTED, "asynchronous calls of this filter are outstanding: mrhip_sync_state before capturing (a replay must find the record they leave)");
    if (!dev_planned && !f->mirror_valid)
        if (int rc = rec_pull(f)) return rc;
    if (x_from && !dev_planned) return fail(MRHIP_ERR_UNSUPPORTED, "a chained call is an asynchronous or captured call");
    if (dev_planned && arb && f->mod_form != 0 && (f->Nphi & (f->Nphi - 1)) != 0)
        return fail(MRHIP_ERR_UNSUPPORTED, "mod form 1 (Julia Base before 0.4) with an N𝜙 that is not a power of two runs the host's serial schedule: no asynchronous or captured calls");
    if (x_len == 0) {                  // nothing to do: zero outputs, history and state unchanged
        f->last_call_dev_planned = false;                      // (no plan kernel ran: the call record still holds an older call's count)
        if (count_dev) MRHIP_CHECK_HIP(hipMemsetAsync(count_dev, 0, sizeof(long long), stream));
        return MRHIP_OK;
    }
    if (!capturing)
        if (int rc = adopt_stream(f, stream)) return rc;
    const int64_t bound = output_bound(f, x_len);
    if (dev_planned) {
        if (y_capacity < bound)
            return fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small: a call planned on the device (mrhip_filt_device_async, HIP-graph capture) needs room for mrhip_outputlength_bound(x_len) outputs");
        if (bound > 0 && !y) return fail(MRHIP_ERR_INVALID_ARG, "y is NULL");
        if (f->nch > 1 && y_stride < bound) return fail(MRHIP_ERR_INVALID_ARG, "y_stride < output bound");
    }

    int64_t n_out = 0;
    bool did_shiftin = false;
    bool sched_inline = false;         // FIRArbitrary / FIRFarrow: this call's schedule ran on the caller's stream although the filter has a schedule stream
    bool hist_in_place = false;        // ... by a filter kernel inside a capture, straight into the slot the replay reads (ShiftFold)
    bool rec_current = false;          // the device record has (or will have, in stream order) this call's end state
    const int hist_next = hist_other(f);
    if (arb) {
        // one range [k0, k0+cnt) of this call's outputs: schedule entries are already in the device buffers
        const size_t yelt = dtype_scalar_size(f->ty) * static_cast<size_t>(f->nc);
        // (d_n, d_acc: the device schedule; n_host: its host copy, or NULL with `spans` = the largest input span of
        //  the aligned tiles of 64 ... 1024 outputs when the schedule is evaluated on the device; dyn: the count comes
        //  from the call record, cnt is its upper bound)
        const void *sched_dn = nullptr, *sched_dacc = nullptr;      // set by the branch that filled them, before its launch_range
        const int *sched_spans = nullptr;
        // whole_call: the range is the call (not a piece of a long one): the kernels that can (arb_pipe_kernel, farrow_wave_kernel) then write the
        // next call's history themselves (ShiftFold) -- inside a capture straight into the slot the replay reads, which takes x_len >= H
        auto launch_range = [&](int64_t k0, int64_t cnt, const int32_t *n_host, const DevCall *dyn, bool whole_call = false) -> int {
            if (!sched_dn || !sched_dacc) return fail(MRHIP_ERR_HIP, "no phase schedule on the device (internal)");
            ShiftFold sf{};
            const bool in_place = capturing;
            if (whole_call && f->H > 0 && !x_from && (!in_place || x_len >= f->H) && MRHIP_ENV_INT("MRHIP_FOLD_SHIFTIN", 1) != 0) {
                sf.hist_new = f->d_hist[in_place ? f->hist_cur : hist_next];
                sf.done = in_place ? f->d_counters + 128 : nullptr;
            }
            void *yk = static_cast<unsigned char *>(y) + static_cast<size_t>(k0) * yelt;
            if (f->kind == MRHIP_FIR_FARROW) {
                FarrowArgs fa{};
                fa.x = x; fa.y = yk; fa.hist = f->d_hist[f->hist_cur]; fa.pnfb = f->d_pnfb;
                fa.n_idx = static_cast<const int *>(sched_dn) + k0; fa.acc = static_cast<const double *>(sched_dacc) + k0;
                fa.x_stride = x_stride; fa.y_stride = y_stride; fa.x_len = x_len; fa.n_out = cnt;
                fa.T = static_cast<int>(f->T); fa.H = static_cast<int>(f->H); fa.polyorder = static_cast<int>(f->polyorder);
                fa.tap_f32 = f->th == MRHIP_F32; fa.nch = static_cast<int>(f->nch);
                fa.seam_below = continuation ? 0 : static_cast<int>(f->T);
                fa.dyn = dyn;
                if (int rc = timing_mark(f, stream)) return rc;
                ArbTileArgs fta;
                size_t flds = 0;
                if (!f->force_generic && f->d_pnfb_t && plan_farrow_wave(fa)) {
                    FarrowArgs fw = fa;
                    fw.pnfb = f->d_pnfb_t;                              // degree-major, padded: [polyorder+1][32]
                    fw.fold = sf;
                    if (sf.hist_new) { did_shiftin = true; hist_in_place = in_place; }
                    MRHIP_CHECK_HIP(launch_farrow_wave(tk, fused, fw, stream, &f->last_kernel, f->num_cus));
                }
                else if (!f->force_generic && plan_farrow_tiled(tk, fa, n_host, sched_spans, f->num_cus, &fta, &flds)) {
                    fta.counters = MRHIP_ENV_INT("MRHIP_PIPE_DYNAMIC", 1) != 0 ? f->d_counters : nullptr;   // (as for FIRArbitrary below)
                    if (sf.hist_new) { fa.fold = sf; did_shiftin = true; hist_in_place = in_place; }
                    MRHIP_CHECK_HIP(launch_farrow_tiled(tk, fused, fa, fta, flds, stream, &f->last_kernel, f->num_cus));
                }
                else
                    MRHIP_CHECK_HIP(launch_farrow(tk, fused, fa, stream, &f->last_kernel));
                return timing_mark(f, stream);
            }
            ArbArgs a{};
            a.x = x; a.y = yk; a.hist = f->d_hist[f->hist_cur]; a.taps = f->d_taps; a.dtaps = f->d_dtaps;
            a.n_idx = static_cast<const int *>(sched_dn) + k0; a.acc = static_cast<const double *>(sched_dacc) + k0;
            a.x_stride = x_stride; a.y_stride = y_stride; a.x_len = x_len; a.n_out = cnt;
            a.T = static_cast<int>(f->T); a.H = static_cast<int>(f->H); a.Nphi = static_cast<int>(f->Nphi);
            a.nch = static_cast<int>(f->nch);
            a.dyn = dyn;
            if (int rc = timing_mark(f, stream)) return rc;
            ArbTileArgs ta;
            size_t lds = 0;
            // (a small call -- at most MRHIP_ARB_SMALL_MAX (150 000) outputs x channels -- runs faster on the universal kernel, one lane per output and
            //  nothing to set up: 1 ch x 1e5 samples 6.7 against 8.6 us, the crossover at 1 ch x 3e5 / 2 ch x 1.5e5; profiles/r05/experiments.md O)
            const bool small_call = cnt * f->nch <= static_cast<int64_t>(MRHIP_ENV_INT("MRHIP_ARB_SMALL_MAX", 150000));
            ArbLaneArgs la;
            const bool lane_ok = !f->force_generic && !small_call && MRHIP_ENV_INT("MRHIP_PIPE_DYNAMIC", 1) != 0;
            hipError_t ew = hipSuccess;
            if (lane_ok && sf.hist_new) a.fold = sf;            // (every kernel below folds shiftin! into its last workgroup)
            if (lane_ok && try_launch_arb_window(tk, fused, a, f->rate, f->d_counters, stream, &f->last_kernel, f->num_cus, &ew)) {
                // (a long call of config 4's shape: one wave per stretch, the window in registers -- kernels_arb_window.hip)
                if (sf.hist_new) { did_shiftin = true; hist_in_place = in_place; }
                MRHIP_CHECK_HIP(ew);
            }
            else if (lane_ok && plan_arb_lane(tk, a, f->rate, &la, &lds)) {
                // (64 channels or more, Float64, a rate >= 1: a lane per channel, the taps in scalar registers -- kernels_arb_lane.hip)
                la.counters = f->d_counters;
                if (sf.hist_new) { a.fold = sf; did_shiftin = true; hist_in_place = in_place; }
                MRHIP_CHECK_HIP(launch_arb_lane(fused, a, la, lds, stream, &f->last_kernel, f->num_cus));
            }
            else if (!f->force_generic && !small_call && plan_arb_tiled(tk, a, n_host, sched_spans, f->num_cus, &ta, &lds)) {
                // the pipe kernel's tiles are handed out from a counter of the filter (its launches are stream-ordered: one at a
                // time); MRHIP_PIPE_DYNAMIC=0: every workgroup takes every gridDim-th tile
                ta.counters = MRHIP_ENV_INT("MRHIP_PIPE_DYNAMIC", 1) != 0 ? f->d_counters : nullptr;
                if (sf.hist_new) { a.fold = sf; did_shiftin = true; hist_in_place = in_place; }
                MRHIP_CHECK_HIP(launch_arb_tiled(tk, fused, a, ta, lds, stream, &f->last_kernel, f->num_cus));
            }
            else {
                if (sf.hist_new) { a.fold = sf; did_shiftin = true; hist_in_place = in_place; }
                MRHIP_CHECK_HIP(launch_arb_generic(tk, fused, a, stream, &f->last_kernel));
            }
            return timing_mark(f, stream);
        };
        ArbState st;
        bool host_loop = false;            // this call's schedule came from the host's serial loop (not from sched_enqueue)
        // Upper bound of the output count (the reference's outputlength estimate, Filters.jl:375-381, + 2 for the
        // rounding of its Float64 recurrence); without the host's copy of the state: for inputDeficit = 1.
        const int64_t est = dev_planned ? bound
            : (x_len >= f->inputDeficit ? static_cast<int64_t>(std::ceil(static_cast<double>(x_len - f->inputDeficit + 1) * f->rate)) + 2 : 0);
        const bool cached = !dev_planned && f->sched_cached && f->sched_xlen == x_len && f->sched_acc0 == f->phiAcc && f->sched_deficit0 == f->inputDeficit;
        static const int64_t piece = [] { const char *v = std::getenv("MRHIP_SCHED_PIECE"); return v && *v ? std::atoll(v) : 262144LL; }();
        if (dev_planned || (!cached && sched_wants_device(f, est))) {
            // The schedule is evaluated on the device (arb_schedule.hip), in front of ONE filter launch that takes the
            // output count from the call record: nothing waits in the middle of the call.  A caller that wants the count
            // (the reference's filt! returns it) collects it from the pinned mirror once the schedule's last kernel has
            // run -- by then the filter kernel is already queued behind it.
            if (est >= 0x7fffffffLL) return fail(MRHIP_ERR_INVALID_ARG, "call too long for one launch (internal)");
            int spans[kSchedSpanSizes];
            span_bounds(f->rate, spans);
            // The schedule runs on the filter's schedule stream, BESIDE the filter kernel of the call before (its inputs are
            // the record, which moved on with that call's FINISH kernel, and nothing else); schedule buffers and call records
            // alternate, events order writer and reader of each (mrhip_filter.h: s_sched).  Inside a capture: one stream.
            // (a chained call's schedule reads the previous stage's call record, which is written on the caller's stream: it runs there)
            // (a call of one small piece -- at most MRHIP_SCHED_INLINE_MAX outputs, default 65 536 -- is launch-bound: its three kernels go down ONE
            //  queue without the three event operations the second stream costs the host; profiles/r05/experiments.md O)
            // (calls arb_lane_kernel will serve: that kernel's persistent workgroups hold every vector register of the chip -- the next call's schedule
            //  cannot run BESIDE it: its tables / chain workgroups either wait for the kernel's end anyway or, when the race at the start lets
            //  them in, both run much longer -- config 4 on a continuing stream read 3.9 or 4.5-5.1 ms per call.  Behind it, on the caller's
            //  stream: 3.93 + 0.2 ms, every call.  profiles/r06/experiments.md I; MRHIP_SCHED_BESIDE_LANE=1: as before)
            const bool lane_shape = f->kind == MRHIP_FIR_ARBITRARY && tk.x_f64 && tk.r_f64 && !tk.complex_x && (f->T == 32 || f->T == 16) && f->rate >= 1.0 &&
                                    f->nch >= 48 && MRHIP_ENV_INT("MRHIP_ARB_LANE", 1) != 0 && MRHIP_ENV_INT("MRHIP_SCHED_BESIDE_LANE", 0) == 0;
            const bool inline_sched = est <= MRHIP_ENV_INT("MRHIP_SCHED_INLINE_MAX", 65536) || lane_shape;
            hipStream_t ss = capturing || !f->s_sched || x_from || inline_sched ? stream : f->s_sched;
            // a chained call's schedule ran on the CALLER's stream (it reads the previous stage's call record there) and wrote the record,
            // the piece states and the path tables: a schedule on the filter's own schedule stream must come behind it
            if (ss == f->s_sched && f->chain_pending)
                if (int rc = chain_event_now(f)) return rc;
            if (ss == f->s_sched) f->sched_dirty = true;
            else if (!capturing && f->s_sched && f->sched_dirty) {        // (mrhip_filter.h: sched_dirty)
                MRHIP_CHECK_HIP(hipEventRecord(f->ev_sdirty, f->s_sched));
                MRHIP_CHECK_HIP(hipStreamWaitEvent(stream, f->ev_sdirty, 0));
                f->sched_dirty = false;
            }
            sched_inline = ss == stream && !capturing && f->s_sched != nullptr;
            SchedOut so{};
            so.buf = capturing ? 0 : f->flip;
            if (!capturing) f->flip ^= 1;
            if (ss != stream && f->ev_filt_valid[so.buf]) MRHIP_CHECK_HIP(hipStreamWaitEvent(ss, f->ev_filt[so.buf], 0));
            if (int rc = sched_enqueue(f, x_len, est, dev_planned ? y_capacity : INT64_MAX, count_dev, !dev_planned, ss, &so, x_from)) return rc;   // (a call that is waited for checks the count against the room itself)
            const int b = so.buf;                                         // (a memo hit names the buffer that holds the entries)
            sched_dn = f->ds_n[b]; sched_dacc = f->ds_acc[b]; sched_spans = spans;
            auto join = [&]() -> int {                                  // the filter kernel behind its schedule
                if (ss == stream) return MRHIP_OK;
                MRHIP_CHECK_HIP(hipEventRecord(f->ev_fin[b], ss));
                MRHIP_CHECK_HIP(hipStreamWaitEvent(stream, f->ev_fin[b], 0));
                return MRHIP_OK;
            };
            const bool room = y && y_capacity >= est && (f->nch == 1 || y_stride >= est);
            bool launched = false;
            if (so.pending && room && est > 0) {
                if (int rc = join()) return rc;
                if (int rc = launch_range(0, est, nullptr, f->d_calls[b], true)) return rc;
                launched = true;
            }
            rec_current = so.pending;
            if (dev_planned) {
                // nobody collects: the host's copy of the state is stale from here on
                f->mirror_valid = false;
                f->last_call_dev_planned = so.pending && launched;        // (a chained call may follow: its input length is this call's count)
                f->last_call_rec = f->d_calls[b];
                if (!capturing) f->async_pending = true;
                if (!capturing && so.pending && !so.periodic && !x_from) {
                    // The host's LOWER BOUND of the drift baseline (it sizes the next calls' pieces: at most 16 x the baseline) moves on
                    // by the fewest outputs this call can have -- xIdx starts at most 1/rate + 1 in and advances at most 1/rate + 1 a
                    // step -- so that a stream of asynchronous calls is not cut into pieces of 4096 forever; mrhip_sync_state brings
                    // the exact value.
                    const double lb = std::floor((static_cast<double>(x_len) - 1.0 / f->rate - 3.0) * f->rate) - 2.0;
                    if (lb > 0.0) f->sched_ksteps += lb;
                }
                n_out = -1;
                st = ArbState{f->phiAcc, f->phiIdx, f->alpha, f->xIdx, f->inputDeficit};
            } else {
                f->last_call_dev_planned = false;
                for (;;) {
                    bool relaunch = false;
                    if (int rc = sched_collect(f, x_len, est, INT64_MAX, count_dev, ss, &so, &relaunch)) return rc;
                    if (!relaunch) break;
                    rec_current = so.pending;
                    if (so.pending && room) {
                        if (int rc = join()) return rc;
                        if (int rc = launch_range(0, est, nullptr, f->d_calls[b], true)) return rc;
                    } else launched = false;
                }
                n_out = so.count;
                st = so.end;
                // (rec_current: the FINISH kernel wrote the record; a call the host evaluated itself pushes it below)
                if (n_out > y_capacity) {
                    if (rec_current) {      // the record moved on with the schedule: take the stream back to the call's start
                        f->sched_cached = false;
                        (void)rec_push(f, ss);
                    }
                    return fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small");
                }
                if (n_out > 0 && !launched) {
                    if (!y) return fail(MRHIP_ERR_INVALID_ARG, "y is NULL");
                    if (f->nch > 1 && y_stride < n_out) return fail(MRHIP_ERR_INVALID_ARG, "y_stride < output count");
                    if (int rc = join()) return rc;
                    if (int rc = launch_range(0, n_out, nullptr, nullptr, true)) return rc;
                }
                f->sched_drift = so.drift; f->sched_ksteps = so.ksteps;
                if (so.periodic || f->per_valid) f->per_pos = so.per_pos_end;
                // this call becomes the memo entry of buffer b (arb_schedule.hip): its entries stay there until b is rewritten
                if (!so.memo_hit) {
                    f->memo_valid = true; f->memo_buf = b;
                    f->memo_acc0 = so.memo_acc0; f->memo_d0 = so.memo_d0; f->memo_xlen = x_len;
                    f->memo_count = n_out; f->memo_end = st; f->memo_drift = so.drift; f->memo_ksteps = so.ksteps;
                    f->memo_per_pos_end = so.per_pos_end;
                }
            }
            if (ss != stream) {                                          // buffer b's next writer waits for this reader
                MRHIP_CHECK_HIP(hipEventRecord(f->ev_filt[b], stream));
                f->ev_filt_valid[b] = true;
            } else if (!capturing && f->s_sched) {                       // (see chain_pending above; also orders buffer b's next writer)
                f->chain_stream = stream;
                f->chain_pending = true;
            }
        } else if (!cached && est > 2 * piece && y && y_capacity >= est && (f->nch == 1 || y_stride >= est) && est < 0x7fffffffLL) {
            host_loop = true;
            if (f->sched_in_flight) { MRHIP_CHECK_HIP(hipEventSynchronize(f->sched_copied)); f->sched_in_flight = false; }
            if (int rc = ensure_sched_capacity(f, static_cast<size_t>(est))) return rc;
            sched_dn = f->d_sched_n; sched_dacc = f->d_sched_acc;      // (the buffers may just have been (re)allocated)
            st = ArbState{f->phiAcc, f->phiIdx, f->alpha, f->inputDeficit, f->inputDeficit};   // xIdx starts at inputDeficit (:715)
            bool done = false;
            int64_t k0 = 0;
            static const bool prof = [] { const char *v = std::getenv("MRHIP_DEBUG"); return v && v[0] == '2'; }();
            double t_rec = 0, t_copy = 0, t_launch = 0;
            auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            // an error in the middle of the pipeline: uploads from the pinned staging may still be in flight, so mark
            // them (the next call waits before it reuses the staging) and report how far the launches got; the
            // filter state is not advanced
            auto bail = [&](int rc) -> int {
                if (hipEventRecord(f->sched_copied, stream) == hipSuccess) f->sched_in_flight = true;
                else { (void)hipGetLastError(); (void)hipStreamSynchronize(stream); }
                if (n_written) *n_written = k0;
                return rc;
            };
            while (!done) {
                int32_t *pn = static_cast<int32_t *>(f->pin_n) + k0;
                double *pa = static_cast<double *>(f->pin_acc) + k0;
                const int64_t room = std::min<int64_t>(piece, est - k0);
                if (room <= 0) return bail(fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small"));   // cannot happen: est is an upper bound
                const double t0 = prof ? now() : 0;
                const int64_t cnt = run_arbitrary_schedule_piece(st, f->delta, f->Nphi, x_len, pn, pa, room, &done, f->mod_form);
                const double t1 = prof ? now() : 0;
                if (cnt > 0) {
                    if (hipMemcpyAsync(static_cast<int32_t *>(f->d_sched_n) + k0, pn, static_cast<size_t>(cnt) * sizeof(int32_t), hipMemcpyHostToDevice, stream) != hipSuccess ||
                        hipMemcpyAsync(static_cast<double *>(f->d_sched_acc) + k0, pa, static_cast<size_t>(cnt) * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess)
                        return bail(fail(MRHIP_ERR_HIP, "uploading the phase schedule failed"));
                    const double t2 = prof ? now() : 0;
                    if (int rc = launch_range(k0, cnt, pn, nullptr)) return bail(rc);
                    if (prof) { t_rec += t1 - t0; t_copy += t2 - t1; t_launch += now() - t2; }
                    k0 += cnt;
                }
            }
            if (prof) std::fprintf(stderr, "[mrhip] arbitrary schedule: recurrence %.2f ms, memcpy enqueue %.2f ms, plan+launch %.2f ms (%lld outputs)\n",
                                   t_rec * 1e3, t_copy * 1e3, t_launch * 1e3, static_cast<long long>(k0));
            MRHIP_CHECK_HIP(hipEventRecord(f->sched_copied, stream));
            f->sched_in_flight = true;
            n_out = k0;
        } else {
        host_loop = true;
        n_out = arb_schedule(f, x_len, &st);   // update(::FIRFarrow), Filters.jl:780-788, is the same recurrence
        if (f->sched_in_flight) { MRHIP_CHECK_HIP(hipEventSynchronize(f->sched_copied)); f->sched_in_flight = false; }
        if (n_out > y_capacity) return fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small");
        if (n_out > 0) {
            if (!y) return fail(MRHIP_ERR_INVALID_ARG, "y is NULL");
            if (f->nch > 1 && y_stride < n_out) return fail(MRHIP_ERR_INVALID_ARG, "y_stride < output count");
            if (int rc = ensure_sched_capacity(f, static_cast<size_t>(n_out))) return rc;
            sched_dn = f->d_sched_n; sched_dacc = f->d_sched_acc;
            std::memcpy(f->pin_n, f->sched_n.data(), static_cast<size_t>(n_out) * sizeof(int32_t));
            std::memcpy(f->pin_acc, f->sched_acc.data(), static_cast<size_t>(n_out) * sizeof(double));
            MRHIP_CHECK_HIP(hipMemcpyAsync(f->d_sched_n, f->pin_n, static_cast<size_t>(n_out) * sizeof(int32_t), hipMemcpyHostToDevice, stream));
            MRHIP_CHECK_HIP(hipMemcpyAsync(f->d_sched_acc, f->pin_acc, static_cast<size_t>(n_out) * sizeof(double), hipMemcpyHostToDevice, stream));
            MRHIP_CHECK_HIP(hipEventRecord(f->sched_copied, stream));
            f->sched_in_flight = true;
            if (int rc = launch_range(0, n_out, f->sched_n.data(), nullptr, true)) return rc;
        }
        }
        // commit the post-call state (Filters.jl:731-735)
        if (!dev_planned) {
            if (host_loop && f->per_valid) {
                // the host's loop evaluated this call (a short one): a cycle found earlier moves on by its outputs -- left where it
                // was, the next ASYNCHRONOUS call would plan from a stale cycle position and be refused by its plan kernel
                // (tests/stress_random.py --async-mix, seed 111: rate 2.5, N𝜙 = 10)
                if (f->per_acc[static_cast<size_t>(f->per_pos)] == f->phiAcc) f->per_pos = (f->per_pos + n_out) % f->per_Q;
                else f->per_valid = false;
            }
            f->phiAcc = st.acc; f->phiIdx = st.phiIdx; f->alpha = st.alpha; f->xIdx = st.xIdx;
            f->inputDeficit = st.inputDeficit;
        }
        f->sched_cached = false;
    } else if (dev_planned) {
        PolyArgs a{};
        a.x = x; a.y = y; a.hist = f->d_hist[f->hist_cur]; a.hist_new = f->d_hist[hist_next]; a.taps = f->d_taps;
        a.x_stride = x_stride; a.y_stride = y_stride; a.x_len = x_len;
        a.n_out = std::max<int64_t>(bound, 1);                          // upper bounds: the plan kernel leaves the call's own values in the call record
        a.u0 = 0; a.d0 = 1;
        a.zero_start_below = continuation ? 0
                           : f->kind == MRHIP_FIR_STANDARD ? f->hLen + 1
                           : f->kind == MRHIP_FIR_DECIMATOR ? f->hLen : 0;
        a.L = static_cast<int>(f->L); a.M = static_cast<int>(f->M);
        a.T = static_cast<int>(f->T); a.H = static_cast<int>(f->H);
        a.nch = static_cast<int>(f->nch);
        a.rec = f->d_rec; a.dyn = f->d_call;
        if (int rc = timing_mark(f, stream)) return rc;
        {
            const hipError_t e = launch_poly_dyn(f, tk, fused, a, x_len, y_capacity, count_dev, stream, &f->last_kernel, &did_shiftin, x_from);
            if (e == hipErrorNotSupported && x_from)
                return fail(MRHIP_ERR_UNSUPPORTED, "a chained call (input length = the previous stage's count, on the device) needs a filter the pair kernels serve (rational_opair_kernel / fir_stream_kernel)");
            MRHIP_CHECK_HIP(e);
        }
        if (int rc = timing_mark(f, stream)) return rc;
        rec_current = true;
        if (x_from) {
            n_out = -1;                 // (the host never learns a chained call's length: its view of the state is re-read from the device)
        } else if (f->mirror_valid || capturing) {
            // the shadow of ONE execution: exact while the host knew the state when the capture (or the run of
            // asynchronous calls) began; re-read from the device before it is next used either way
            const CallPlan p = plan_rational(f->kind, f->L, f->M, f->phiIdx, f->inputDeficit, x_len);
            n_out = p.n_out;
            f->phiIdx = p.phi_end;
            f->inputDeficit = p.d_end;
        } else {
            n_out = -1;
        }
        f->mirror_valid = false;
        f->last_call_dev_planned = true;
        f->last_call_rec = f->d_call;
        if (!capturing) f->async_pending = true;
    } else {
        f->last_call_dev_planned = false;
        const CallPlan p = plan_rational(f->kind, f->L, f->M, f->phiIdx, f->inputDeficit, x_len);
        n_out = p.n_out;
        // reference: error() before any work, Filters.jl:460 (Standard), :503 (Interpolator), :550 (Rational)
        if (n_out > y_capacity) return fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small");
        if (n_out > 0) {
            if (!y) return fail(MRHIP_ERR_INVALID_ARG, "y is NULL");
            if (f->nch > 1 && y_stride < n_out) return fail(MRHIP_ERR_INVALID_ARG, "y_stride < output count");
            PolyArgs a{};
            a.x = x; a.y = y; a.hist = f->d_hist[f->hist_cur]; a.hist_new = f->d_hist[hist_next]; a.taps = f->d_taps;
            a.x_stride = x_stride; a.y_stride = y_stride; a.x_len = x_len; a.n_out = n_out;
            a.u0 = p.phi0 - 1; a.d0 = p.d0;
            a.zero_start_below = continuation ? 0
                               : f->kind == MRHIP_FIR_STANDARD ? f->hLen + 1
                               : f->kind == MRHIP_FIR_DECIMATOR ? f->hLen : 0;
            a.L = static_cast<int>(f->L); a.M = static_cast<int>(f->M);
            a.T = static_cast<int>(f->T); a.H = static_cast<int>(f->H);
            a.nch = static_cast<int>(f->nch);
            a.rec = f->d_rec; a.dyn = nullptr; a.phi_end = p.phi_end; a.d_end = p.d_end;
            if (int rc = timing_mark(f, stream)) return rc;
            MRHIP_CHECK_HIP(launch_poly(f, tk, fused, a, stream, &f->last_kernel, &did_shiftin, f->d_counters, &rec_current));
            if (int rc = timing_mark(f, stream)) return rc;
        }
        f->phiIdx = p.phi_end;
        f->inputDeficit = p.d_end;
    }
    // the device record follows every call in stream order: a call whose kernels did not file its end state pushes it
    // (FIRStandard / FIRInterpolator have no state to carry, but the record's count and call counter -- what mrhip_sync_state returns --
    //  follow every call of every kind)
    if (!rec_current)
    {
        // (every write of a FIRArbitrary / FIRFarrow record in program order: on the schedule stream, behind whatever the caller's stream wrote
        //  last -- or, when this call's schedule ran on the caller's stream, there, with the schedule stream's next user behind it)
        const bool on_sched = arb && f->s_sched && !capturing && !sched_inline;
        if (on_sched) { if (int rc = sched_stream_behind_chain(f)) return rc; f->sched_dirty = true; }
        if (int rc = rec_push(f, on_sched ? f->s_sched : stream, -1, std::max<int64_t>(n_out, 0))) return rc;
        if (on_sched) f->async_pending = true;       // (nobody waits for that push: a capture must not start before it has run)
        if (sched_inline) {
            f->chain_stream = stream;
            f->chain_pending = true;
            f->async_pending = true;
        }
    }

    // history <- last H samples of [history ; x]   (shiftin!, support.jl:61-80), ping-pong buffers
    if (f->H > 0 && !did_shiftin) {
        HistArgs ha{};
        ha.x = x; ha.hist_old = f->d_hist[f->hist_cur]; ha.hist_new = f->d_hist[hist_next];
        ha.x_stride = x_stride; ha.x_len = x_len; ha.H = static_cast<int>(f->H); ha.nch = static_cast<int>(f->nch);
        ha.dyn = x_from ? f->last_call_rec : nullptr;          // (a chained call: the length its plan / FINISH kernel took from the previous stage)
        MRHIP_CHECK_HIP(launch_shiftin(tk, ha, stream));
    }
    if (f->H > 0) {
        if (capturing && hist_in_place) {
            // (the filter kernel wrote the slot the replay reads)
        } else if (capturing) {
            // a replay reads the slot baked into the node: bring the new history back into it instead of moving on
            MRHIP_CHECK_HIP(hipMemcpyAsync(f->d_hist[f->hist_cur], f->d_hist[hist_next],
                                           static_cast<size_t>(f->nch) * f->H * x_elt(f), hipMemcpyDeviceToDevice, stream));
        } else {
            f->hist_cur = hist_next;
        }
    }
    if (capturing) f->captured = true;
    if (n_written) *n_written = n_out;
    return MRHIP_OK;
}

// filt!(buffer, self, x) on device memory.  The reference takes a Vector of any length (Int64 indices); a launch indexes
// 31 bits, and a FIRArbitrary / FIRFarrow launch holds its phase schedule in device memory, so a long call is cut into
// launch-sized pieces here -- chunked == unchunked bit for bit (logical window, closed-form / carried state), see
// filt_device_one for the one place where a piece must know it is not the start of the call.
static int filt_device_any(mrhip_filter *f, const void *x, int64_t x_len, int64_t x_stride, void *y,
                           int64_t y_capacity, int64_t y_stride, int64_t *n_written, void *stream, bool async, long long *count_dev)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    if (n_written) *n_written = 0;
    if (f->ring_open) return fail(MRHIP_ERR_INVALID_ARG, "the filter feeds a ring (mrhip_ring_open): push chunks into the ring, or close it first");
    const bool arb = f->kind == MRHIP_FIR_ARBITRARY || f->kind == MRHIP_FIR_FARROW;
    // samples per launch: inputs AND outputs stay below 2^31 (FIRInterpolator and FIRRational with L > M write more than
    // they read); at most 2^24 schedule entries per FIRArbitrary launch
    const int64_t env_max = MRHIP_ENV_INT("MRHIP_LAUNCH_MAX", 0);              // tests: force the split at small sizes
    int64_t step = env_max > 0 ? env_max : (1LL << 30);
    if (env_max <= 0 && (f->kind == MRHIP_FIR_INTERPOLATOR || f->kind == MRHIP_FIR_RATIONAL) && f->L > f->M)
        step = std::max<int64_t>(static_cast<int64_t>((static_cast<__int128>(step) * f->M) / f->L), 1);
    if (f->kind == MRHIP_FIR_INTERPOLATOR && env_max > 0) step = std::max<int64_t>(step / f->L, 1);
    if (arb && env_max <= 0) step = std::max<int64_t>(4096, std::min<int64_t>(step, static_cast<int64_t>(static_cast<double>(1LL << 24) / f->rate)));
    if (x_len <= step) return filt_device_one(f, x, x_len, x_stride, y, y_capacity, y_stride, n_written, stream, false, async, count_dev);
    if (x_len < 0 || y_capacity < 0) return fail(MRHIP_ERR_INVALID_ARG, "negative length");
    if (!x) return fail(MRHIP_ERR_INVALID_ARG, "x is NULL");
    if (async || stream_is_capturing(static_cast<hipStream_t>(stream)))
        return fail(MRHIP_ERR_UNSUPPORTED, "a call of this length is issued in several launches whose places in y depend on the counts of the earlier ones: neither capturable nor asynchronous");
    if (!f->mirror_valid)
        if (int rc = rec_pull(f)) return rc;
    if (!arb) {   // reference: error() before any work, Filters.jl:460 (Standard), :503 (Interpolator), :550 (Rational)
        const int64_t total = plan_rational(f->kind, f->L, f->M, f->phiIdx, f->inputDeficit, x_len).n_out;
        if (total > y_capacity) return fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small");
        if (total > 0 && !y) return fail(MRHIP_ERR_INVALID_ARG, "y is NULL");
        if (f->nch > 1 && y_stride < total) return fail(MRHIP_ERR_INVALID_ARG, "y_stride < output count");
    }
    const size_t xelt = dtype_scalar_size(f->tx) * static_cast<size_t>(f->nc);
    const size_t yelt = dtype_scalar_size(f->ty) * static_cast<size_t>(f->nc);
    int64_t k = 0;
    for (int64_t a = 0; a < x_len; a += step) {
        const int64_t len = std::min<int64_t>(step, x_len - a);
        int64_t got = 0;
        const int rc = filt_device_one(f, static_cast<const unsigned char *>(x) + static_cast<size_t>(a) * xelt, len, x_stride,
                                       static_cast<unsigned char *>(y) + static_cast<size_t>(k) * yelt, y_capacity - k, y_stride, &got, stream, a > 0);
        if (rc) { if (n_written) *n_written = k; return rc; }
        k += got;
    }
    if (n_written) *n_written = k;
    return MRHIP_OK;
}

int mrhip_filt_device(mrhip_filter *f, const void *x, int64_t x_len, int64_t x_stride, void *y,
                      int64_t y_capacity, int64_t y_stride, int64_t *n_written, void *stream)
{
    return filt_device_any(f, x, x_len, x_stride, y, y_capacity, y_stride, n_written, stream, false, nullptr);
}

int mrhip_filt_device_async(mrhip_filter *f, const void *x, int64_t x_len, int64_t x_stride, void *y,
                            int64_t y_capacity, int64_t y_stride, int64_t *count_out, void *stream)
{
    static_assert(sizeof(long long) == sizeof(int64_t), "the count is written by the device as a long long");
    return filt_device_any(f, x, x_len, x_stride, y, y_capacity, y_stride, nullptr, stream, true, reinterpret_cast<long long *>(count_out));
}

// A CHAINED asynchronous call: this filter's input is what `prev`'s latest device-planned call (asynchronous or captured, earlier
// on the same stream) wrote -- `prev`'s count, known on the device only, is this call's input length; x_len_bound is its upper
// bound (mrhip_outputlength_bound of `prev`'s input), for which the launch is sized and y must have room.
int mrhip_filt_device_chained(mrhip_filter *f, const mrhip_filter *prev, const void *x, int64_t x_len_bound, int64_t x_stride, void *y,
                              int64_t y_capacity, int64_t y_stride, int64_t *count_out, void *stream)
{
    if (!f || !prev) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    if (f->device != prev->device) return fail(MRHIP_ERR_INVALID_ARG, "the two filters live on different devices");
    if (!prev->last_call_dev_planned || !prev->last_call_rec)
        return fail(MRHIP_ERR_INVALID_ARG, "the previous filter's latest call was not planned on the device (mrhip_filt_device_async, or a call under capture): its count is not in its call record");
    if (x_len_bound >= (1LL << 30)) return fail(MRHIP_ERR_UNSUPPORTED, "a chained call is one launch");
    if ((f->kind == MRHIP_FIR_ARBITRARY || f->kind == MRHIP_FIR_FARROW) && static_cast<double>(x_len_bound) * f->rate >= static_cast<double>(1LL << 24))
        return fail(MRHIP_ERR_UNSUPPORTED, "a chained call is one launch (2^24 schedule entries for FIRArbitrary / FIRFarrow)");
    return filt_device_one(f, x, x_len_bound, x_stride, y, y_capacity, y_stride, nullptr, stream, false, true,
                           reinterpret_cast<long long *>(count_out), prev->last_call_rec);
}

// SEVERAL INDEPENDENT STREAMS, ONE LAUNCH.  The reference's streaming usage is one FIRFilter per signal (README.md:87-141): N
// signals are N objects with N phases and N call lengths, and a filter object's channels cannot serve them (channels share
// the call length and therefore the state).  One launch per stream cannot fill 256 CUs (a 1e6-sample chunk of one channel
// is all launch ramp: 6-7 % of the HBM roofline); here the streams of one launch are the scheduling groups of the pair
// kernel: every workgroup works for one stream and takes that stream's signal, history, taps, record, lengths and
// call-start state from a descriptor (MultiDesc, pair_loader.h: pair_take_dyn).  The rational family on its two pair kernels
// (FIRRational / FIRInterpolator: rational_opair_kernel, FIRStandard / FIRDecimator: fir_stream_kernel); anything else is the
// plain loop of single calls (same results).
int mrhip_filt_device_multi(mrhip_filter *const *filters, int n, const void *const *x, const int64_t *x_len, void *const *y,
                            const int64_t *y_capacity, int64_t *n_written, void *stream_)
{
    if (!filters || n < 1 || !x || !x_len || !y || !y_capacity) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    mrhip_filter *f0 = filters[0];
    if (!f0) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    auto single_calls = [&]() -> int {
        for (int i = 0; i < n; ++i) {
            int64_t got = 0;
            const int64_t xs = x_len[i], ys = y_capacity[i];
            if (int rc = mrhip_filt_device(filters[i], x[i], xs, xs, y[i], ys, ys, &got, stream_)) return rc;
            if (n_written) n_written[i] = got;
        }
        return MRHIP_OK;
    };
    bool same = !(f0->kind == MRHIP_FIR_ARBITRARY || f0->kind == MRHIP_FIR_FARROW) && !f0->force_generic && n > 1 && n <= 4096;
    for (int i = 0; i < n && same; ++i) {
        const mrhip_filter *f = filters[i];
        if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
        for (int j = 0; j < i; ++j)
            if (filters[j] == f) return fail(MRHIP_ERR_INVALID_ARG, "the same filter twice in one multi-stream call");
        same = f->kind == f0->kind && f->L == f0->L && f->M == f0->M && f->T == f0->T && f->tx == f0->tx && f->th == f0->th &&
               f->numerics == f0->numerics && f->device == f0->device && x_len[i] > 0 && x_len[i] < 0x7fffffffLL && x[i] && y[i];
    }
    if (!same || stream_is_capturing(stream)) return single_calls();
    DeviceGuard guard(f0->device);
    if (!guard.ok) return fail(MRHIP_ERR_HIP, "hipSetDevice failed");
    const TypeKey tk = type_key(f0);
    const bool fused = f0->numerics == MRHIP_NUMERICS_FUSED;
    // what every stream's call will do (closed form), and the geometry of the launch: planned for the longest stream
    std::vector<CallPlan> plans(static_cast<size_t>(n));
    int64_t n_out_max = 0, nch_total = 0, x_len_max = 0;
    for (int i = 0; i < n; ++i) {
        mrhip_filter *f = filters[i];
        if (!f->mirror_valid)
            if (int rc = rec_pull(f)) return rc;
        plans[i] = plan_rational(f->kind, f->L, f->M, f->phiIdx, f->inputDeficit, x_len[i]);
        if (plans[i].n_out > y_capacity[i]) return fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small");
        n_out_max = std::max(n_out_max, plans[i].n_out);
        x_len_max = std::max(x_len_max, x_len[i]);
        nch_total += f->nch;
    }
    if (n_out_max < 1) return single_calls();                        // (short inputs only: history shifts, no kernel worth sharing)
    PolyArgs a{};
    a.x = x[0]; a.y = y[0]; a.hist = f0->d_hist[f0->hist_cur]; a.hist_new = f0->d_hist[hist_other(f0)]; a.taps = f0->d_taps;
    a.x_stride = x_len_max; a.y_stride = n_out_max; a.x_len = x_len_max; a.n_out = n_out_max;
    a.u0 = 0; a.d0 = 1;
    a.zero_start_below = f0->kind == MRHIP_FIR_STANDARD ? f0->hLen + 1 : f0->kind == MRHIP_FIR_DECIMATOR ? f0->hLen : 0;   // support.jl:46 (per call: every stream's call starts here)
    a.L = static_cast<int>(f0->L); a.M = static_cast<int>(f0->M); a.T = static_cast<int>(f0->T); a.H = static_cast<int>(f0->H);
    a.nch = static_cast<int>(std::min<int64_t>(nch_total, 0x7fffffff));
    PairArgs pa;
    dim3 block;
    size_t lds = 0;
    const bool column = f0->L == 1;                                  // FIRStandard / FIRDecimator: the streaming single-column kernel
    if (!(column ? plan_fir_stream(tk, a, f0->num_cus, &pa, &block, &lds) : plan_rational_opair(tk, fused, a, f0->num_cus, &pa, &block, &lds))) return single_calls();
    // descriptors: pinned staging -> device, owned by the first filter of the call
    const size_t bytes = static_cast<size_t>(n) * sizeof(MultiDesc);
    if (int rc = multi_staging(f0, bytes)) return rc;
    MultiDesc *d = static_cast<MultiDesc *>(f0->multi_pin);
    unsigned steps_max = 0;
    for (int i = 0; i < n; ++i) {
        mrhip_filter *f = filters[i];
        if (int rc = adopt_stream(f, stream)) return rc;
        const CallPlan &p = plans[i];
        const int64_t spc = (p.n_out + pa.P - 1) / pa.P;
        if (spc * f->nch >= (1LL << 31)) return fail(MRHIP_ERR_INVALID_ARG, "stream too long for one launch");
        MultiDesc &m = d[i];
        m.x = x[i]; m.y = y[i]; m.hist = f->d_hist[f->hist_cur]; m.hist_new = f->d_hist[hist_other(f)]; m.taps = f->d_taps; m.rec = f->d_rec;
        m.x_stride = x_len[i]; m.y_stride = y_capacity[i]; m.x_len = x_len[i]; m.n_out = p.n_out;
        m.u0 = p.phi0 - 1; m.d0 = p.d0; m.phi_end = p.phi_end; m.d_end = p.d_end;
        m.steps_per_channel = static_cast<unsigned>(spc);
        m.total_steps = static_cast<unsigned>(spc * f->nch);
        m.spc_magic = spc <= 1 ? 0xffffffffu : static_cast<unsigned>((1ULL << 32) / static_cast<unsigned long long>(spc));
        m.nch = static_cast<int>(f->nch);
        m.P_blk = 0; m.q0 = 0;
        steps_max = std::max(steps_max, m.total_steps);
    }
    MRHIP_CHECK_HIP(hipMemcpyAsync(f0->multi_dev, f0->multi_pin, bytes, hipMemcpyHostToDevice, stream));
    MRHIP_CHECK_HIP(hipEventRecord(f0->multi_ev, stream));
    f0->multi_in_flight = true;
    a.rec = f0->d_rec; a.dyn = nullptr;
    a.multi = static_cast<const MultiDesc *>(f0->multi_dev); a.multi_n = n;
    pa.total_steps = steps_max;                                     // (sizes the grid: workgroups per stream <= tiles of the longest)
    if (int rc = timing_mark(f0, stream)) return rc;
    if (column) MRHIP_CHECK_HIP(launch_fir_stream(fused, a, pa, block, lds, stream, &f0->last_kernel, f0->num_cus, f0->d_counters));
    else MRHIP_CHECK_HIP(launch_rational_opair(fused, a, pa, block, lds, stream, &f0->last_kernel, f0->num_cus, f0->d_counters));
    if (int rc = timing_mark(f0, stream)) return rc;
    for (int i = 0; i < n; ++i) {                                    // the kernel filed every stream's end state and history
        mrhip_filter *f = filters[i];
        f->phiIdx = plans[i].phi_end; f->inputDeficit = plans[i].d_end;
        if (plans[i].n_out == 0) {
            // (a stream without outputs in this call has no tiles: its workgroups still shift its history and file its state)
        }
        if (f->H > 0) f->hist_cur = hist_other(f);
        f->last_kernel = f0->last_kernel;
        if (n_written) n_written[i] = plans[i].n_out;
    }
    return MRHIP_OK;
}

int64_t mrhip_outputlength_bound(const mrhip_filter *f, int64_t inputlength)
{
    if (!f || inputlength < 0) return -1;
    return output_bound(f, inputlength);
}

int mrhip_sync_state(mrhip_filter *f, int64_t *last_n_written)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    DeviceGuard guard(f->device);
    if (int rc = rec_pull(f)) return rc;
    const DevStream r = *f->h_rec;
    if (last_n_written) *last_n_written = r.n_written;
    if (r.error != 0) {
        if (int rc = push_state(f)) return rc;                      // (clears the sticky error; the state stays)
        return fail(r.error, r.error == MRHIP_ERR_BUFFER_TOO_SMALL ? "buffer is too small (a device-planned call clipped its outputs)"
                                                                   : "a device-planned call failed (the stream was moved off the cycle its schedule relies on, or its schedule overran the bound)");
    }
    return MRHIP_OK;
}

int mrhip_filt_device_chunked(mrhip_filter *f, const void *x, int64_t x_len, int64_t x_stride, int64_t chunk, void *y,
                              int64_t y_capacity, int64_t y_stride, int64_t *n_written, void *stream)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    if (n_written) *n_written = 0;
    if (chunk < 1) return fail(MRHIP_ERR_INVALID_ARG, "chunk must be >= 1");
    if (x_len < 0 || y_capacity < 0) return fail(MRHIP_ERR_INVALID_ARG, "negative length");
    const size_t xelt = dtype_scalar_size(f->tx) * static_cast<size_t>(f->nc);
    const size_t yelt = dtype_scalar_size(f->ty) * static_cast<size_t>(f->nc);
    // The signal is resident, so for the pfb kernels the chunk loop collapses: the outputs of consecutive filt! calls
    // are the outputs of one call over the concatenation (the dot product runs over the LOGICAL window [history ; x],
    // support.jl:16-31, and the state recurrence is the same closed form, Filters.jl:558-571), bit for bit, and they
    // land in the same places of y.  One launch per LAUNCH_MAX samples instead of one per chunk: a 1e6-sample chunk of
    // one channel (7.7 MB) cannot fill 256 CUs for longer than the launch ramp.  FIRStandard/FIRDecimator keep the
    // per-chunk loop: their seam dot product starts from zero (support.jl:46), which is visible per CALL (the sign of an
    // all-(-0) sum); FIRArbitrary/FIRFarrow keep it because their wall time is the host's serial phase recurrence.
    // MRHIP_CHUNKED_PER_CALL=1 forces the per-chunk loop (measurements of genuinely arriving chunks).
    const bool per_call = MRHIP_ENV_INT("MRHIP_CHUNKED_PER_CALL", 0) == 1;
    int64_t step = chunk;
    if (!per_call && (f->kind == MRHIP_FIR_INTERPOLATOR || f->kind == MRHIP_FIR_RATIONAL) && chunk < x_len) {
        const int64_t total = plan_rational(f->kind, f->L, f->M, f->phiIdx, f->inputDeficit, x_len).n_out;
        // an undersized buffer must fail at the piece the caller's own loop would fail at: leave that to the loop
        if (total <= y_capacity && (f->nch == 1 || y_stride >= total)) {
            const int64_t launch_max = (1LL << 30) / (f->kind == MRHIP_FIR_INTERPOLATOR ? f->L : 1);   // inputs and outputs per launch stay below 2^31
            step = std::max<int64_t>(chunk, launch_max / chunk * chunk);     // whole chunks per launch
        }
    }
    int64_t k = 0;
    for (int64_t a = 0; a < x_len; a += step) {
        const int64_t len = std::min<int64_t>(step, x_len - a);
        int64_t got = 0;
        const int rc = mrhip_filt_device(f, static_cast<const unsigned char *>(x) + static_cast<size_t>(a) * xelt, len, x_stride,
                                         static_cast<unsigned char *>(y) + static_cast<size_t>(k) * yelt, y_capacity - k, y_stride, &got, stream);
        if (rc) { if (n_written) *n_written = k; return rc; }
        k += got;
    }
    if (n_written) *n_written = k;
    return MRHIP_OK;
}

static int ensure_staging(mrhip_filter *f, int slot, size_t xbytes, size_t ybytes)
{
    if (xbytes > f->d_xcap[slot]) {
        if (f->d_xbuf[slot]) (void)hipFree(f->d_xbuf[slot]);
        f->d_xbuf[slot] = nullptr; f->d_xcap[slot] = 0;
        MRHIP_CHECK_HIP(hipMalloc(&f->d_xbuf[slot], xbytes));
        f->d_xcap[slot] = xbytes;
    }
    if (ybytes > f->d_ycap[slot]) {
        if (f->d_ybuf[slot]) (void)hipFree(f->d_ybuf[slot]);
        f->d_ybuf[slot] = nullptr; f->d_ycap[slot] = 0;
        MRHIP_CHECK_HIP(hipMalloc(&f->d_ybuf[slot], ybytes));
        f->d_ycap[slot] = ybytes;
    }
    return MRHIP_OK;
}

// Host-pointer path.  The signal is cut into pieces of 64 MiB of input (MRHIP_HOST_PIECE_KB); piece i is
// copied in on s_in, filtered on own_stream and copied out on s_out, two staging slots each way, so that the H2D copy of
// piece i+1 and the D2H copy of piece i-1 run beside the kernel of piece i and the two PCIe directions are busy at the
// same time.  Chunked == unchunked bit for bit, so the pieces are invisible in the result.  With page-locked caller
// buffers (hipHostMalloc / hipHostRegister) the copies are true DMA at PCIe rate; pageable buffers are staged by the
// runtime and each copy call returns when its staging is done.
int mrhip_filt_host(mrhip_filter *f, const void *x, int64_t x_len, int64_t x_stride, void *y, int64_t y_capacity,
                    int64_t y_stride, int64_t *n_written)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    if (n_written) *n_written = 0;
    if (x_len < 0 || y_capacity < 0) return fail(MRHIP_ERR_INVALID_ARG, "negative length");
    if (x_len == 0) return MRHIP_OK;
    if (!x) return fail(MRHIP_ERR_INVALID_ARG, "x is NULL");
    if (f->nch > 1 && x_stride < x_len) return fail(MRHIP_ERR_INVALID_ARG, "x_stride < x_len");
    const int64_t count = mrhip_next_output_count(f, x_len);
    if (count > y_capacity) return fail(MRHIP_ERR_BUFFER_TOO_SMALL, "buffer is too small");
    if (count > 0 && !y) return fail(MRHIP_ERR_INVALID_ARG, "y is NULL");
    if (f->nch > 1 && y_stride < count) return fail(MRHIP_ERR_INVALID_ARG, "y_stride < output count");

    DeviceGuard guard(f->device);
    if (!guard.ok) return fail(MRHIP_ERR_HIP, "hipSetDevice failed");
    const size_t xe = x_elt(f), ye = y_elt(f);
    // MRHIP_HOST_PIECE_KB (tests, tuning): input KiB per piece, read per call
    const int64_t env_kb = MRHIP_ENV_INT("MRHIP_HOST_PIECE_KB", 0);
    const int64_t piece_kb = env_kb > 0 ? env_kb : 64 * 1024;
    int64_t piece = (piece_kb << 10) / static_cast<int64_t>(xe * static_cast<size_t>(f->nch));
    piece = std::max<int64_t>(piece, 256);
    const bool pipelined = x_len > piece + piece / 2;
    if (!pipelined) piece = x_len;
    if (!f->s_in) {
        MRHIP_CHECK_HIP(hipStreamCreateWithFlags(&f->s_in, hipStreamNonBlocking));
        MRHIP_CHECK_HIP(hipStreamCreateWithFlags(&f->s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->ev_in[i], hipEventDisableTiming));
            MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->ev_k[i], hipEventDisableTiming));
            MRHIP_CHECK_HIP(hipEventCreateWithFlags(&f->ev_out[i], hipEventDisableTiming));
        }
    }
    hipStream_t sk = f->own_stream;
    // per-piece output bound (for sizing the staging): the exact count of the first piece + slack for the others
    const int64_t y_piece_cap = pipelined ? std::max<int64_t>(mrhip_outputlength(f, piece), 0) + 8 + 2 * (f->L / std::max<int64_t>(f->M, 1) + 1) : count;
    const int nslots = pipelined ? 2 : 1;
    for (int sl = 0; sl < nslots; ++sl)
        if (int rc = ensure_staging(f, sl, std::max<size_t>(static_cast<size_t>(piece) * xe * f->nch, 16),
                                    std::max<size_t>(static_cast<size_t>(y_piece_cap) * ye * f->nch, 16))) return rc;
    const size_t x_pitch = static_cast<size_t>(f->nch > 1 ? x_stride : x_len) * xe;
    const size_t y_pitch = static_cast<size_t>(f->nch > 1 ? y_stride : count) * ye;
    int64_t k = 0;
    int it = 0;
    bool used[2] = {false, false};
    // any failure inside the loop: copies that touch the caller's x / y may still be in flight on s_in / s_out, and the
    // pieces already filtered have advanced the filter -- drain the three streams and report how far the call got
    auto bail = [&](int rc) -> int {
        (void)hipStreamSynchronize(f->s_in); (void)hipStreamSynchronize(sk); (void)hipStreamSynchronize(f->s_out);
        (void)hipGetLastError();
        if (n_written) *n_written = k;
        return rc;
    };
#define MRHIP_HOSTPATH_HIP(expr)                                                                          \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return bail(fail(MRHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); \
    } while (0)
    for (int64_t a = 0; a < x_len; a += piece, ++it) {
        const int sl = it & 1;
        const int64_t len = std::min<int64_t>(piece, x_len - a);
        const size_t xrow = static_cast<size_t>(len) * xe;
        // the x slot is free once the kernel of two pieces ago has run; the y slot once its D2H copy has
        if (used[sl]) MRHIP_HOSTPATH_HIP(hipStreamWaitEvent(f->s_in, f->ev_k[sl], 0));
        MRHIP_HOSTPATH_HIP(hipMemcpy2DAsync(f->d_xbuf[sl], xrow, static_cast<const unsigned char *>(x) + static_cast<size_t>(a) * xe, x_pitch, xrow,
                                         static_cast<size_t>(f->nch), hipMemcpyHostToDevice, f->s_in));
        MRHIP_HOSTPATH_HIP(hipEventRecord(f->ev_in[sl], f->s_in));
        MRHIP_HOSTPATH_HIP(hipStreamWaitEvent(sk, f->ev_in[sl], 0));
        if (used[sl]) MRHIP_HOSTPATH_HIP(hipStreamWaitEvent(sk, f->ev_out[sl], 0));
        const int64_t cap = std::min<int64_t>(y_piece_cap, count - k);
        int64_t nw = 0;
        // (an undersized slot cannot happen: y_piece_cap bounds every piece; the whole call was checked against y_capacity)
        if (int rc = filt_device_one(f, f->d_xbuf[sl], len, len, f->d_ybuf[sl], cap, std::max<int64_t>(cap, 1), &nw, sk, a > 0)) {
            return bail(rc);
        }
        MRHIP_HOSTPATH_HIP(hipEventRecord(f->ev_k[sl], sk));
        if (nw > 0) {
            MRHIP_HOSTPATH_HIP(hipStreamWaitEvent(f->s_out, f->ev_k[sl], 0));
            const size_t yrow = static_cast<size_t>(nw) * ye;
            MRHIP_HOSTPATH_HIP(hipMemcpy2DAsync(static_cast<unsigned char *>(y) + static_cast<size_t>(k) * ye, y_pitch, f->d_ybuf[sl],
                                             static_cast<size_t>(std::max<int64_t>(cap, 1)) * ye, yrow, static_cast<size_t>(f->nch), hipMemcpyDeviceToHost, f->s_out));
        }
        MRHIP_HOSTPATH_HIP(hipEventRecord(f->ev_out[sl], f->s_out));
        used[sl] = true;
        k += nw;
    }
#undef MRHIP_HOSTPATH_HIP
    MRHIP_CHECK_HIP(hipStreamSynchronize(f->s_out));
    MRHIP_CHECK_HIP(hipStreamSynchronize(sk));
    if (n_written) *n_written = k;
    return MRHIP_OK;
}

int mrhip_synchronize(mrhip_filter *f, void *stream)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    DeviceGuard guard(f->device);
    MRHIP_CHECK_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return MRHIP_OK;
}

int mrhip_filt_once(const void *h, int64_t hLen, int th, int64_t num, int64_t den, double rate, int64_t Nphi,
                    const void *x, int64_t x_len, int tx, void *y, int64_t y_capacity, int64_t *n_written, int device)
{
    mrhip_filter *f = nullptr;
    int rc = rate > 0.0 ? mrhip_create_arbitrary(h, hLen, th, rate, Nphi, tx, 1, device, &f)
                        : mrhip_create_rational(h, hLen, th, num, den, tx, 1, device, &f);
    if (rc) return rc;
    rc = mrhip_filt_host(f, x, x_len, x_len, y, y_capacity, y_capacity, n_written);
    mrhip_destroy(f);
    return rc;
}

int mrhip_set_timing(mrhip_filter *f, int enabled)
{
    if (!f) return fail(MRHIP_ERR_INVALID_ARG, "NULL filter");
    f->timing = enabled != 0;
    f->timing_stride = enabled > 1 ? enabled : 1;     // enabled = n > 1: bracket every n-th compute launch
    f->timing_group = enabled < -1 ? -enabled : 1;    // enabled = -n < -1: one bracket around every n consecutive launches
    f->timing_launch = 0; f->timing_open = false; f->ev_skip = false;
    f->ev_used = 0;
    return MRHIP_OK;
}

int mrhip_timing_read(mrhip_filter *f, int64_t *n_launches, double *total_ms)
{
    if (!f || !n_launches || !total_ms) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    DeviceGuard guard(f->device);
    *n_launches = 0;
    *total_ms = 0.0;
    const size_t pairs = f->ev_used / 2;              // (group mode: a start event without its stop is dropped)
    for (size_t i = 0; i < pairs; ++i) {
        float ms = 0.f;
        MRHIP_CHECK_HIP(hipEventSynchronize(f->ev_pool[2 * i + 1]));
        MRHIP_CHECK_HIP(hipEventElapsedTime(&ms, f->ev_pool[2 * i], f->ev_pool[2 * i + 1]));
        *total_ms += ms;
    }
    *n_launches = static_cast<int64_t>(pairs) * f->timing_group;
    f->ev_used = 0;
    f->timing_launch = 0; f->timing_open = false;
    return MRHIP_OK;
}

const char *mrhip_last_kernel_name(const mrhip_filter *f) { return f ? f->last_kernel : ""; }

int mrhip_schedule_info(const mrhip_filter *f, int64_t *info, int n)
{
    if (!f || !info || n < 0) return fail(MRHIP_ERR_INVALID_ARG, "NULL argument");
    if (f->kind != MRHIP_FIR_ARBITRARY && f->kind != MRHIP_FIR_FARROW) return fail(MRHIP_ERR_INVALID_ARG, "not a FIRArbitrary / FIRFarrow filter");
    const int64_t v[9] = {f->splan.ok, f->splan.ncand, f->splan.nwin, f->per_valid ? f->per_Q : 0,
                          f->stat_host_steps, f->stat_periodic_steps, f->stat_device_pieces, f->stat_fallback_pieces, f->stat_memo_hits};
    for (int i = 0; i < n && i < 9; ++i) info[i] = v[i];
    return MRHIP_OK;
}

}  // extern "C"

#include "ring_api.inc"
