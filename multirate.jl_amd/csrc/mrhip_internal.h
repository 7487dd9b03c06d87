// mrhip_internal.h -- shared declarations of libmultirate_hip.so (not installed).
//
// Layering inside csrc/:
//   host_logic.cpp      data-independent bookkeeping (taps2pfb, outputlength, closed-form state
//                       advance, FIRArbitrary phase schedule).  No HIP calls.
//   kernels_*.hip       gfx950 device kernels + their launchers.
//   api.hip             the extern "C" surface declared in include/multirate_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "multirate_hip.h"

namespace mrhip {

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define MRHIP_CHECK_HIP(expr)                                                                  \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::mrhip::fail(MRHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

inline bool dtype_is_complex(int dt) { return dt == MRHIP_C64 || dt == MRHIP_C128; }
inline bool dtype_is_f64(int dt) { return dt == MRHIP_F64 || dt == MRHIP_C128; }
inline size_t dtype_scalar_size(int dt) { return dtype_is_f64(dt) ? 8 : 4; }
inline size_t dtype_size(int dt) { return dtype_scalar_size(dt) * (dtype_is_complex(dt) ? 2 : 1); }

// ---------------------------------------------------------------------------------------
// host logic (host_logic.cpp)
// ---------------------------------------------------------------------------------------
int64_t taps2pfb(const void *h, int64_t hLen, int th, int64_t Nphi, void *out);
int64_t nextphase(int64_t phase, int64_t L, int64_t M);
int64_t outputlength_ratio(int64_t inputlength, int64_t L, int64_t M, int64_t initialPhi);
int64_t inputlength_ratio(int64_t outputlength, int64_t L, int64_t M, int64_t initialPhi);

// What one filt call on the rational family will do, from the call-start state alone
// (SURVEY.md Appendix A "Per-call view"; reference loop: src/Filters.jl:558-571).
struct CallPlan {
    int64_t n_out = 0;         // outputs this call (per channel)
    int64_t phi0 = 1, d0 = 1;  // call-start (phiIdx, inputDeficit), 1-based
    int64_t phi_end = 1;       // state after the call
    int64_t d_end = 1;
    bool short_input = false;  // xLen < inputDeficit  (Filters.jl:543-547)
};
CallPlan plan_rational(int kind, int64_t L, int64_t M, int64_t phiIdx, int64_t inputDeficit, int64_t xLen);

// FIRArbitrary phase recurrence (src/Filters.jl:663-673, 715-734) run on the host.
struct ArbState {
    double acc = 1.0;   // 𝜙Accumulator
    int64_t phiIdx = 1;
    double alpha = 0.0;
    int64_t xIdx = 1;
    int64_t inputDeficit = 1;
};
// Runs the schedule for one call.  If n_idx/acc_out are non-null they receive, per output, the
// 1-based input index and the phase accumulator value used (phiIdx = floor(acc), alpha = acc - phiIdx).
// Returns the number of outputs; `st` is advanced to the post-call state.
// mod_form != 0: mod() of update() (Filters.jl:668) as rem(y + rem(x, y), y), the form of Julia Base before 0.4 (mrhip_set_mod_form)
int64_t run_arbitrary_schedule(ArbState &st, double delta, int64_t Nphi, int64_t xLen,
                               std::vector<int32_t> *n_idx, std::vector<double> *acc_out, int mod_form = 0);
int64_t run_arbitrary_schedule_piece(ArbState &st, double delta, int64_t Nphi, int64_t xLen, int32_t *n_idx, double *acc_out,
                                     int64_t max_outputs, bool *done, int mod_form = 0);

// ---------------------------------------------------------------------------------------
// FIRArbitrary / FIRFarrow phase schedule on the device (kernels_schedule.hip; model: scripts/sched_model.py)
// ---------------------------------------------------------------------------------------
constexpr int kSchedSeg = 64;                 // steps per segment
constexpr int kSchedGroup = 64 * kSchedSeg;   // steps per group (one workgroup): pieces are whole groups
constexpr int kSchedMaxWin = 64;              // candidates per segment the tables kernel has LDS for
constexpr int kSchedNoFail = 0x7fffffff;
constexpr int kSchedSpanSizes = 5;            // tile sizes (kSchedSpanBase << i outputs: 64 ... 1024) whose largest input span the emit kernel reports
constexpr int kSchedSpanBase = 64;
inline int sched_span_index(long long tile_out) { int z = 0; while (z + 1 < kSchedSpanSizes && (static_cast<long long>(kSchedSpanBase) << z) < tile_out) ++z; return z; }

struct SchedPlan {           // constants of one (delta, Nphi)
    double delta, N, invN;
    double umin, inv_umin;   // grid of the values the recurrence can produce
    double G, inv_G;         // shifts by multiples of G commute with every rounding of the recurrence
    double halfwin;          // (nwin / 2) * umin
    int ncand, nwin;         // G / umin; candidates per segment
    int pow2;                // Nphi is a power of two (the quotient by N is then an exact scaling)
    int ok;                  // 0: outside what the device evaluation covers (the host loop runs instead)
};
struct SchedPieceState {     // true state at the start of a piece + the running drift estimate
    double acc;
    long long xIdx;
    double drift;            // sum over verified pieces of (true end - un-rounded end), phase units
    double ksteps;           // steps those pieces covered
};
struct SchedStatus {
    int fail_piece;          // kSchedNoFail, or the first piece whose verification failed
    int done;                // != 0: the call's end (first xIdx > x_len) was found, in piece done - 1
    long long end_k;         // number of outputs of the call
    double end_acc;          // state after the last output's update()
    long long end_xIdx;
    int max_span[kSchedSpanSizes];   // largest n[last] - n[first] over aligned tiles of kSchedSpanBase << i outputs
    int groups_done;         // workgroups of the LAST piece's emit kernel that are through: the last one is the call's FINISH
    int pad[2];
    long long x_len;         // the call's input length as the BEGIN kernel resolved it (a chained call: the previous stage's count)
    double end_drift;        // drift baseline at the call's end: the piece's start baseline + the deviation measured up to end_k
    double end_ksteps;       //   (the next call continues from there: a stream of one-piece calls builds its baseline too)
};
struct SchedGroupEntry { double shift; int advance; int next; };        // map of one group: candidate -> (next, +shift, +xIdx)
struct SchedGroupStart { double shift; long long advance; int cand; int pad; };   // true start of a group, as candidate + shift
struct SchedPieceArgs {
    SchedPieceState *state;  // [pieces + 1]
    SchedStatus *status;
    double *pathT;           // [segments of the piece][nwin]: start value of the segment when the group starts at candidate c0
    int *pathW;              //                                xIdx advance from the group's start
    SchedGroupEntry *gtab;   // [groups][nwin]
    SchedGroupStart *gstart; // [groups]
    int *sched_n;            // the call's schedule buffers
    double *sched_acc;
    long long k0;            // output number of the piece's first step within the call
    long long x_len;
    int piece, ngroups;
    int corrupt_group;       // test hook: -1, or the group whose table is falsified (MRHIP_SCHED_CORRUPT)
};
SchedPlan make_sched_plan(double delta, int64_t Nphi, int win_mult = 1, int win_min = 4);
struct SchedFuseArgs;
// fu != NULL: the call's BEGIN rides in the tables kernel of piece 0 and / or its FINISH in the emit kernel of the last piece
hipError_t launch_schedule_piece(const SchedPlan &c, const SchedPieceArgs &a, hipStream_t s, const SchedFuseArgs *fu = nullptr);
// The two one-lane kernels round the pieces of a call (kernels_schedule.hip).  BEGIN arms the status word and writes the
// call-start state of piece 0 -- from the device record, or from the host's values when it just evaluated a prefix itself.
// FINISH turns the pieces' verdict into the call's result: output count into the DevCall the filter kernel reads (clamped
// to the caller's room in y), end state and drift estimate into the record and its pinned mirror.  A piece that did not
// verify is either reported (the host redoes it and continues: the call is waited for anyway) or, for a call nobody
// waits for, redone on the spot by the serial recurrence, from that piece's verified start state to the end of the call.
struct DevStream;
struct DevCall;
struct SchedBeginArgs {
    DevStream *rec;
    SchedStatus *status;
    SchedPieceState *state;
    int use_host;                 // 1: (acc, xIdx, drift, ksteps) below are the state at schedule entry k_first
    double acc; long long xIdx; double drift, ksteps;
    const DevCall *x_from;        // a chained call: the input length is this record's count (the x_len argument: its upper bound)
};
struct SchedFinishArgs {
    DevStream *rec, *mirror;
    DevCall *call;
    SchedStatus *status;
    SchedPieceState *state;       // [pieces + 1]
    SchedPieceState *fail_state;  // pinned: the verified start state of a piece that failed (for the host's redo)
    int *sched_n; double *sched_acc;
    long long *count_out;         // optional, device-accessible: receives the call's output count
    long long k_first;            // schedule entry the first piece starts at (> 0 behind a host prefix)
    double ks_first;              // the drift baseline the host sized the pieces with (piece sizes are recomputed from it)
    long long pmax, est, x_len, y_capacity;
    int np;                       // pieces enqueued
    int serial_fallback;          // redo a failed piece (and everything behind it) here instead of reporting it
};
// BEGIN and FINISH without launches of their own (a call of one piece: four dependent launches instead of six before the filter
// kernel).  begin: every workgroup of piece 0's tables kernel derives the call-start state itself, workgroup 0 files status
// and state[0] for the kernels behind it.  finish: the workgroups of the last piece's emit kernel count themselves off
// (SchedStatus::groups_done, release / acquire at device scope); the last one through is the FINISH kernel.
struct SchedFuseArgs {
    int begin, finish;
    int fold_chain;               // no chain kernel for this piece (few groups): every emit workgroup walks the groups' maps to its own start
    SchedBeginArgs b;
    long long b_x_len, b_k_first;
    SchedFinishArgs f;
};
hipError_t launch_sched_begin(const SchedBeginArgs &a, long long x_len, long long k_first, hipStream_t s);
hipError_t launch_sched_finish(const SchedPlan &c, const SchedFinishArgs &a, hipStream_t s);
// piece sizes of a call: piece i covers steps [k, k + P), P the largest power-of-two multiple of a group that is at most
// pmax and at most 16x the drift baseline behind it (one rule for the host that enqueues and the kernel that sums up)
__host__ __device__ inline long long sched_piece_steps(double ks, long long pmax)
{
    long long P = kSchedGroup;
    while (P * 2 <= pmax && static_cast<double>(P * 2) <= 16.0 * ks) P *= 2;
    return P;
}
double sched_anchor_host(const SchedPlan &c, double acc_p, double k);   // un-rounded phase after k steps (host_logic.cpp)

// ---------------------------------------------------------------------------------------
// the stream state ON THE DEVICE (stream_state.hip)
// ---------------------------------------------------------------------------------------
// The reference mutates 𝜙Idx / inputDeficit / 𝜙Accumulator at the end of every filt! (src/Filters.jl:571-572, 627-628,
// 734-735, update() :663-673).  Here that state has a device-resident record per filter, kept current IN STREAM ORDER by
// every call: a call the host planned writes the end state it computed (carried in the kernel arguments: one lane of the
// filter kernel, or stream_set_kernel); a DEVICE-PLANNED call (mrhip_filt_device_async, every call captured into a HIP
// graph, every FIRArbitrary / FIRFarrow call whose schedule the device evaluates) reads the record in a one-lane plan
// kernel, leaves what its filter kernel needs in a DevCall, and advances the record there -- so a replayed graph continues
// the stream by itself, and the host learns counts and state from a pinned mirror of the record, never by waiting in the
// middle of a call.
struct DevStream {
    long long phiIdx, inputDeficit;   // 1-based, reference field names
    double acc;                       // 𝜙Accumulator
    double drift, ksteps;             // FIRArbitrary / FIRFarrow: running drift estimate of the device schedule
    long long per_pos;                // ... position in a detected cycle of the accumulator
    long long n_written;              // outputs per channel of the last call
    long long calls;                  // filt! calls completed on this stream
    long long fallback_steps;         // schedule steps redone by the serial loop on the device
    int error;                        // sticky until read: mrhip_status of a device-planned call (buffer too small, ...)
    int sched_fail;                   // the last call's first schedule piece that did not verify (kSchedNoFail: none)
};
struct DevCall {                      // what the kernels of ONE device-planned call read
    long long n_out;                  // outputs per channel
    long long u0, d0;                 // rational family: phi0 - 1, inputDeficit at call start
    unsigned steps_per_channel, total_steps, spc_magic, pad0;   // pair kernels: the step walk for this n_out
    long long per_pos, per_xbase;     // PERIODIC schedule: cycle position / x offset at the call's first entry
    long long k_done;                 // schedule entries valid so far (a call whose schedule is continued after a host redo)
    long long x_len;                  // rational family: the call's input length (a chained call takes it from the previous stage's count)
};
// Closed form of the rational loop (host_logic.cpp: plan_rational), usable on both sides
struct CallPlanPOD { long long n_out, phi0, d0, phi_end, d_end; int short_input; };
__host__ __device__ inline CallPlanPOD plan_rational_pod(int kind, long long L, long long M, long long phiIdx, long long inputDeficit, long long xLen)
{
    CallPlanPOD p{0, 1, 1, 1, 1, 0};
    if (kind == MRHIP_FIR_STANDARD) { p.n_out = xLen; return p; }
    if (kind == MRHIP_FIR_INTERPOLATOR) { p.n_out = L * xLen; return p; }
    p.phi0 = kind == MRHIP_FIR_DECIMATOR ? 1 : phiIdx;
    p.d0 = inputDeficit;
    if (xLen < inputDeficit) { p.short_input = 1; p.n_out = 0; p.phi_end = phiIdx; p.d_end = inputDeficit - xLen; return p; }
    const long long a = (xLen - inputDeficit + 1) * L - p.phi0 + 1;          // outputlength_ratio, Filters.jl:352-357
    p.n_out = a >= 0 ? (a + M - 1) / M : -((-a) / M);
    const long long u_end = (p.phi0 - 1) + p.n_out * M;
    p.phi_end = u_end % L + 1;
    p.d_end = p.d0 + u_end / L - xLen;
    return p;
}

// ---------------------------------------------------------------------------------------
// device-side parameter blocks
// ---------------------------------------------------------------------------------------
struct PolyArgs {            // rational family: STANDARD / DECIMATOR / INTERPOLATOR / RATIONAL
    const void *x;           // device, planar [ch][x_stride]
    void *y;                 // device, planar [ch][y_stride]
    const void *hist;        // device, [ch][H] samples of Tx (call-start history)
    void *hist_new;          // device, [ch][H]: the other ping-pong buffer; a kernel that performs shiftin! itself
                             // (support.jl:61-80) writes the call-end history here and reports it (*did_shiftin)
    const void *taps;        // device, R-typed, [Nphi][T] (column = phase, oldest-sample tap first)
    long long x_stride, y_stride;
    long long x_len;         // samples per channel in this call
    long long n_out;         // outputs per channel in this call
    long long u0;            // phi0 - 1
    long long d0;            // inputDeficit at call start (1-based)
    long long zero_start_below;  // outputs whose 1-based input index n < this start from +0
                                 // (support.jl:46; STANDARD: hLen+1, DECIMATOR: hLen, else 0)
    int L, M, T, H;
    int nch;
    // the device-resident stream state (see DevStream).  dyn == NULL: the host planned this call; `rec` (if set) receives
    // the end state (phi_end, d_end) and n_out from one lane of the kernel.  dyn != NULL: n_out, u0, d0 (and the step walk of
    // the pair kernels) are read from *dyn, which the call's plan kernel filled; the values above are upper bounds.
    DevStream *rec;
    const DevCall *dyn;
    long long phi_end, d_end;
    // SEVERAL INDEPENDENT STREAMS in one launch (mrhip_filt_device_multi; pair kernels only): multi != NULL: workgroup b works
    // for stream b % multi_n alone and takes everything above that differs from stream to stream -- signal, history, taps,
    // record, lengths, call-start state, step walk -- from multi[b % multi_n]; the values above are those of the launch's
    // planning (the longest stream).
    const struct MultiDesc *multi;
    int multi_n;
    // THE RESIDENT RING CONSUMER (ring.hip; pair kernels' RING instantiations only): ring_dev != NULL: the launch is the consumer
    // of a ring of arriving chunks -- signal, output, lengths, call-start state come per chunk from RingDesc records, `hist` is
    // the ring's array of history slots; everything else above describes the filter (taps, L, M, T, H, nch).
    struct RingDev *ring_dev;
    struct RingHost *ring_host;
};

struct MultiDesc {           // one independent stream (one FIRFilter of the reference: README.md:87-141) of a multi-stream launch
    const void *x;
    void *y;
    const void *hist;
    void *hist_new;
    const void *taps;
    DevStream *rec;
    long long x_stride, y_stride, x_len, n_out, u0, d0, phi_end, d_end;
    unsigned steps_per_channel, total_steps, spc_magic;
    int nch;
    // PERIOD BLOCKS (rational_opair_kernel for L > 512, kernels_rational_opair.hip: plan_rational_opair_blocks): the "stream" is one
    // block of P_blk consecutive outputs of every period of 2L; q0 = first sample of the block's sub-range of a step's input
    int P_blk, q0;
};

// ---------------------------------------------------------------------------------------
// the resident ring consumer (ring.hip, pair_loader.h: pair_ring_loader_wave)
// ---------------------------------------------------------------------------------------
// The reference's streaming usage is a loop of filt(self, chunk) calls on one stateful FIRFilter (README.md:87-141).  One launch per
// arriving 1e6-sample chunk of one channel is all launch latency (13 us per call round an 8 us kernel that could take 1.5).  Here
// ONE resident kernel consumes chunk descriptors the host pushes into a ring in pinned memory: a feeder wave copies them into
// device memory, every other workgroup takes grabs of J steps by ticket (ticket t -> workgroup t mod G) across chunk boundaries,
// so several chunks are in flight at once; the stream state is planned by the host per chunk (closed form, Filters.jl:558-571),
// the history (shiftin!, support.jl:61-80) travels between chunks through history slots in device memory, and the host learns
// of a chunk's completion from a flag in pinned memory.  Every wait in the kernel has a deadline: it ends by itself when the
// ring stays empty for `idle_ticks`.
constexpr int kRingDepth = 64;            // descriptor / history slots; at most kRingDepth - 1 chunks in flight
constexpr int kRingTraceRows = 512, kRingTraceTiles = 32;   // diagnostics: RingDev::trace
constexpr int kRingShards = 32;           // completion counters per chunk: a grab counts on shard (ticket mod 32), each on a line of its own
struct RingDesc {                         // one arriving chunk: 16 quad-words
    unsigned long long x, y;              // device addresses
    long long x_stride, y_stride, x_len, n_out;
    long long u0, d0;                     // call-start state: phi0 - 1, inputDeficit
    // the KEY (quad-words 8 and 9: one aligned 16-byte unit, written and read by ONE instruction each): whoever reads a key whose
    // seq_lo names the chunk it expects in this slot, and whose tickets hold a grab that is still to be done, reads a descriptor
    // that cannot be under rewrite (a slot is recycled only when its chunk is complete)
    unsigned long long tile_base;         // tickets [tile_base, tile_base + ngrabs) are this chunk's grabs
    unsigned ngrabs, seq_lo;              // seq_lo: low half of the chunk number (~0: the feeder is rewriting the slot)
    unsigned steps_per_channel, total_steps;
    unsigned spc_magic, flags;            // flags bit 0: the outputs of this (small) chunk are stored write-through and its completion raised at once (RingDev::flush_req)
    unsigned long long seq;               // chunk number (host's copy; the device goes by the key)
    unsigned long long pad[3];
};
static_assert(sizeof(RingDesc) == 128, "RingDesc is 16 quad-words");
constexpr int kRingKeyQword = 8;          // index of the key in quad-words (16-byte aligned)
struct RingHost {                         // pinned host memory: host writes head / close / desc, device writes done / stopped
    unsigned long long head;              // chunks published
    unsigned long long close;             // != 0: consume what is published, then leave
    unsigned long long stopped;           // written by the feeder when the kernel leaves: 1 closed, 2 idle deadline, 3 a wait ran into its deadline
    unsigned long long arrived, grid;     // written by the feeder while the kernel starts: workgroups on the chip so far, of how many (ring_launch)
    unsigned long long pad[3];
    unsigned long long done[kRingDepth];  // done[seq % depth] = seq + 1 once chunk seq is complete (outputs written through)
    RingDesc desc[kRingDepth];
};
struct RingDev {                          // device memory
    unsigned long long head;              // chunks whose descriptors are in desc[]
    unsigned long long closed;            // 0 open; 1 closed by the host; 2 idle deadline; 3 error (a wait ran into its deadline)
    unsigned long long idle_ticks;        // deadline of every wait, in 100 MHz ticks
    unsigned long long opts;              // experiments (MRHIP_RING_OPTS): bit 0 no descriptor prefetch, bit 1 grabs reported at their own end
    unsigned long long pad[4];
    unsigned long long hist_seq[kRingDepth];   // hist_seq[s % depth] == s + 1: slot s % depth holds chunk s's call-start history
    unsigned chunk_done[kRingDepth];      // SHARDS of chunk (slot) complete
    // grabs of chunk (slot) completed, counted per shard: ONE counter per chunk took every grab's atomic add on one address -- 13 ns each, in
    // series: 174 grabs of a 1e6-sample chunk = 2.3 us per chunk, which WAS the ring's rate (profiles/r05/experiments.md)
    unsigned shard_done[kRingDepth][kRingShards][32];
    // diagnostics (MRHIP_RING_OPTS bit 8 = 256), 100 MHz ticks summed over the waves that report: [0] compute waves: whole life, [1] ... at the
    // tile barrier, [2] ... draining stores before a report, [3] compute waves reporting; [4] loader waves: whole life, [5] ... polling for
    // chunks, [6] ... in stage_tile + descriptor prefetch (issue and landing), [7] ... at the tile barrier, [8] loader waves reporting,
    // [9] tiles, [10] FLUSH tiles, [11] ... waiting for a history slot, [12] ... in find_chunk with descriptors at hand
    // [13] feeder batches, [14] ... ticks from seeing `head` move to having published, [15] descriptors; per loader: [16] idle tiles BEFORE its
    // last real tile, [17] / [18] min / max over loaders of the time of the last real tile, [19] max of the first, [20] / [21] min / max real tiles
    unsigned long long stats[24];
    // Completion by ONE L2 write-back per XCD and chunk (the default; MRHIP_RING_OPTS bit 13 = 8192: write-through output stores instead, the round's
    // first form -- a fifth slower, experiments.md Q, R).  The compute waves store plainly; the workgroup that reports a chunk's last grab (every
    // grab's stores are in its XCD's L2 by then) files REQUEST r = ++flush_req with the chunk's number; every loader wave looks at flush_req once
    // per tile, and the first of an XCD to see a request its XCD has not taken on (flush_claim) writes that XCD's L2 back (buffer_wbl2) and
    // raises flush_done; whoever then finds every XCD past requests (flush_pub, F] raises the host's flags of those chunks, in request order.
    unsigned long long arrived;             // workgroups of the resident kernel that have started (the host checks that ALL did: ring_launch)
    unsigned long long grid;                // ... of how many (written by the feeder)
    unsigned long long flush_req;
    unsigned long long flush_pub;
    unsigned long long flush_pad[12];
    unsigned long long flush_ent[kRingDepth][2];       // request r at [r % depth]: (r, chunk number), one 16-byte store
    unsigned long long flush_claim[8][16];             // per XCD, a line of its own
    unsigned long long flush_done[8][16];
    // experiment (MRHIP_RING_OPTS bit 10): tickets HANDED OUT instead of dealt -- eight queues (ticket t in queue t mod 8 at position t div 8),
    // one per group of workgroups that share an XCD; next_ticket[q][0] = positions of queue q taken so far.  Slower than dealing (experiments.md L).
    unsigned long long next_ticket[8][16];
    __attribute__((aligned(128))) RingDesc desc[kRingDepth];
    // diagnostics (MRHIP_RING_OPTS bit 8): per workgroup, (time a real tile's staging began, time it was published, ticket) of its first
    // kRingTraceTiles real tiles; row 0: the feeder's (time, head) pairs.  Dumped to the file MRHIP_RING_TRACE names when the kernel has left.
    unsigned long long trace[kRingTraceRows][3 * kRingTraceTiles];
};

// shiftin! (support.jl:61-80) folded into a FIRArbitrary / FIRFarrow filter kernel: the workgroup that leaves LAST writes the next call's
// history -- one launch less per call (small calls are launch-bound: DESIGN.md §9), and inside a stream capture no copy node either:
// there `hist_new` is the slot `hist` itself (everybody else is through with it; the host folds only when x_len >= H, so the new history
// comes from x alone).
struct ShiftFold {
    void *hist_new;          // NULL: not folded (a launch of shiftin_kernel follows)
    unsigned *done;          // workgroups through; zero between launches (the last workgroup re-arms it).  NULL: hist_new is the OTHER buffer --
                             // nothing to wait for, the grid's last workgroup copies
};

struct ArbArgs {             // FIRArbitrary
    const void *x;
    void *y;
    const void *hist;
    const void *taps;        // pfb  [Nphi][T]
    const void *dtaps;       // dpfb [Nphi][T]
    const int *n_idx;        // device, per output: 1-based input index
    const double *acc;       // device, per output: phase accumulator
    long long x_stride, y_stride;
    long long x_len;
    long long n_out;
    int T, H, Nphi;
    int nch;
    const DevCall *dyn;      // != NULL: n_out is read from it (a device-planned call; the value above is an upper bound)
    ShiftFold fold;          // arb_pipe_kernel, arb_tiled_kernel, arb_generic_kernel
};

struct FarrowArgs {          // FIRFarrow
    const void *x;
    void *y;
    const void *hist;
    const double *pnfb;      // device, [T][polyorder+1] ascending powers (values representable in Th)
    const int *n_idx;        // device, per output: 1-based input index
    const double *acc;       // device, per output: the Float64 phase 𝜙Idx
    long long x_stride, y_stride;
    long long x_len;
    long long n_out;
    int T, H, polyorder;
    int tap_f32;             // currentTaps is a Vector{Float32}: round every evaluated tap to Float32
    int nch;
    int seam_below;          // outputs whose 1-based input index n < this start from +0 (support.jl:46): T, or 0 for a
                             // piece that continues a call (mrhip_filt_device splits long calls)
    const DevCall *dyn;      // != NULL: n_out is read from it
    ShiftFold fold;          // farrow_wave_kernel, farrow_pipe_kernel, farrow_tiled_kernel
};

struct HistArgs {            // shiftin! (src/support.jl:61-80) for every channel
    const void *x;
    const void *hist_old;
    void *hist_new;
    long long x_stride;
    long long x_len;
    int H;
    int nch;
    const DevCall *dyn;      // a chained device-planned call: the input length is dyn->x_len (x_len above: its upper bound)
};

struct TileArgs {            // tiling of the phase-stationary kernel (kernels_phase_stationary.hip)
    int c;                   // P = c*L active lanes per workgroup
    int P;
    int J;                   // outputs per lane per tile
    int tile_len;            // samples staged in LDS per tile (per copy)
    int copyB_offset_bytes;  // byte offset of the shifted copy inside a stage (4-byte samples only)
    int stage_bytes;         // LDS bytes per pipeline stage (two stages)
    int dma_rounds;          // LDS-DMA instructions per wave, per copy, per tile
    int ablate;              // timing experiments only (MRHIP_PS_ABLATE): bit0 skip staging, bit1 skip stores
    int x_aligned16;         // channel bases allow 16-byte vector loads
    int by_position;         // lane map: 0 = one lane per output, 1 = one lane per input position
    long long tiles_per_channel;
    long long total_tiles;
};

struct PairArgs {            // tiling of the pair-per-lane rational kernel (kernels_rational_pair.hip)
    int c;                   // period = c*M input positions = c*L outputs per step
    int P;                   // c*L
    int cM;                  // c*M
    int J;                   // steps per tile
    int tile_len;            // samples staged per tile (multiple of 4)
    int tail;                // samples a tile of j steps needs beyond j*cM (window overhang; pair_loader.h)
    int dma_rounds;          // LDS-DMA instructions per wave per tile
    int stage_bytes;         // LDS bytes per pipeline stage
    int ns;                  // pipeline stages (the DMA runs ns-1 tiles ahead of the compute waves)
    int nc;                  // components per sample: 1 = real, 2 = complex
    int x_f64, r_f64;        // Float64 samples / Float64 arithmetic (opair kernel)
    int ablate;              // timing experiments only (MRHIP_PS_ABLATE)
    unsigned steps_per_channel;   // ceil(n_out / P)
    unsigned total_steps;         // steps_per_channel * channels
    unsigned steps_per_group;     // ceil(total_steps / ngroups): group g owns steps [g*S, (g+1)*S)
    int ngroups;                  // scheduling groups; workgroup b draws grabs of J steps from group b % ngroups
    int grid_cap;                 // ring mode: at most that many workgroups (0: as many as the occupancy query says the chip holds)
    int flags_off;                // LDS byte offset of the per-stage tile descriptors
    int pad_every;                // > 0: one 16-byte pad chunk after every pad_every data chunks of a staged tile (pair_loader.h)
    int bank_off;                 // LDS byte offset of the tap bank staged by the compute waves before the first tile (-1: none;
                                  // the loader wave then joins one extra barrier before its first tile barrier)
    int static_grabs;             // 1: grabs are dealt round-robin without atomics (small launches)
    int rt;                       // fir_stream: 1 = the run-time-decimation kernel (kernels_fir_stream_rt.hip), rt_rd = bytes per LDS read
    int rt_rd;
    unsigned *counters;           // device: [g*64] next grab of group g, [ngroups*64] workgroups finished (re-arms all)
    unsigned spc_magic;           // floor(2^32 / steps_per_channel) (0xffffffff for 1): step number -> channel by multiply-high
    // steps of a workgroup are Sout outputs apart in y and lds_step samples apart in LDS (P and cM unless the workgroup works on a
    // BLOCK of the period: L > 512); run_chunks > 0: a tile is staged as one run of run_chunks 16-byte chunks PER STEP, cM samples apart
    // in the signal and packed in LDS (run_magic = ceil(2^32 / run_chunks)); q0: first sample of the block's runs within a step's input
    int Sout, lds_step, run_chunks, q0;
    unsigned run_magic;
    long long o0;            // d0 - T: x index of LDS sample 0 of tile 0 (negative => history)
    long long tile_in;       // J*c*M
    long long tile_out;      // J*c*L
    long long tiles_per_channel;
    long long total_tiles;
    unsigned long long *probe;   // diagnostics (MRHIP_PAIR_PROBE=1): per-workgroup (shader cycles, 100 MHz ticks) of the tile loop
};

// scheduling counters of the pair kernels: [g*64] for group g < 32 (256 bytes apart), [32*64] workgroups finished
constexpr size_t kCounterBytes = 33 * 256;

struct ArbTileArgs {         // tiling of the FIRArbitrary kernel (kernels_arbitrary.hip)
    int cpl;                 // channels per lane (1, 2 or 4): a tile covers cpl channels
    int tap_pitch;           // elements between PFB columns in LDS (T + 1)
    int bank_elems;          // elements per tap bank in LDS
    int x_offset_bytes;      // byte offset of the sample tile in LDS
    int max_span;            // samples the largest tile touches
    int prefetch;            // arb_tiled_kernel: the next tile's samples are loaded into registers a tile ahead
    int copyb_pad;           // samples between the end of sample copy A and the start of copy B (bank stagger)
    int pipe;                // 1: arb_pipe_kernel (kernels_arb_pipe.hip): one copy, two sample buffers, tiles of 256 outputs
    int row_pitch;           // pipe kernels with LDS-DMA staging (prefetch = 1): samples between the rows of a tile (whole 16-byte chunks)
    int dma_slots;           //   ... and 1 KiB wave transfers per copy of the tile
    long long tile_out;      // outputs per tile
    long long tiles_per_channel;
    long long total_tiles;
    unsigned *counters;      // pipe kernels: [0] RUNS of run_tiles tiles handed out beyond the first grid-full, [64] workgroups
                             // through (both zero between launches: the last workgroup re-arms them); NULL: tile += gridDim (static)
    int run_tiles;           // consecutive tiles per hand-out (>= 2)
};

// A device-planned FIRArbitrary / FIRFarrow call (ArbArgs::dyn / FarrowArgs::dyn): the output count comes from the call
// record (the schedule's FINISH kernel left it there) and the tiling follows it; the launch was sized with upper bounds.
// `ngroups` = channel groups per stretch of outputs (tiles that share one stretch of the schedule).
__device__ __forceinline__ void tiles_take_dyn(long long &n_out, ArbTileArgs &ta, long long &ngroups, const DevCall *dyn)
{
    ngroups = ta.tiles_per_channel > 0 ? ta.total_tiles / ta.tiles_per_channel : 1;
    if (dyn) {
        n_out = dyn->n_out;
        ta.tiles_per_channel = (n_out + ta.tile_out - 1) / ta.tile_out;
        ta.total_tiles = ta.tiles_per_channel * ngroups;
    }
}

// Every compute kernel is launched through launch_kernel().  Optional timing mode MRHIP_TIMING_ATTACH=1
// (api.hip:timing_mark): a start/stop event pair armed by api.hip is attached to the next kernel's own dispatch
// (hipExtLaunchKernel) instead of being recorded around it.
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
extern thread_local LaunchEvents g_launch_events;

template <typename F, typename... Args>
inline void launch_kernel(F kfn, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args)
{
    const LaunchEvents ev = g_launch_events;
    g_launch_events = LaunchEvents{};
    if (ev.start && ev.stop)
        hipExtLaunchKernelGGL(kfn, grid, block, static_cast<std::uint32_t>(lds), s, ev.start, ev.stop, 0u, args...);
    else
        hipLaunchKernelGGL(kfn, grid, block, lds, s, args...);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) + hipOccupancyMaxActiveBlocksPerMultiprocessor, remembered per
// (kernel, device, block size, LDS bytes): a streaming launch must not pay for them every time (api.hip).
hipError_t occupancy_cached(const void *kfn, unsigned block, size_t lds, int *per_cu);

// Tuning / test knobs from the environment: read ONCE per process and call site -- a launch must not pay for getenv.
// MRHIP_ENV_DYNAMIC=1 (set by tests/conftest.py: tests switch kernels between calls) re-reads them on every use.
inline int env_int_read(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}
inline bool env_dynamic()
{
    static const bool d = env_int_read("MRHIP_ENV_DYNAMIC", 0) != 0;
    return d;
}
#define MRHIP_ENV_INT(name, dflt) ([]() -> int { static const int v_ = ::mrhip::env_int_read(name, dflt); \
                                                 return ::mrhip::env_dynamic() ? ::mrhip::env_int_read(name, dflt) : v_; }())

// dtype combination a kernel is instantiated for
struct TypeKey {
    bool x_f64;      // Tx scalar is double
    bool r_f64;      // compute/output scalar is double
    bool complex_x;  // NC == 2
};

// ---------------------------------------------------------------------------------------
// launchers (kernels_*.hip).  Each returns hipSuccess or the launch error and writes the kernel
// name it dispatched into *kname.
// ---------------------------------------------------------------------------------------
hipError_t launch_poly_generic(const TypeKey &tk, bool fused, const PolyArgs &a, hipStream_t s, const char **kname);
hipError_t launch_arb_generic(const TypeKey &tk, bool fused, const ArbArgs &a, hipStream_t s, const char **kname);
hipError_t launch_farrow(const TypeKey &tk, bool fused, const FarrowArgs &a, hipStream_t s, const char **kname);
bool plan_farrow_tiled(const TypeKey &tk, const FarrowArgs &a, const int32_t *n_idx_host, const int *spans, int num_cus, ArbTileArgs *out, size_t *lds);
hipError_t launch_farrow_tiled(const TypeKey &tk, bool fused, const FarrowArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                               const char **kname, int num_cus);
hipError_t launch_shiftin(const TypeKey &tk, const HistArgs &a, hipStream_t s);
// least-squares polynomial fit of y[0..n) at x = 1..n (support.jl:85-88); coef receives polyorder+1 ascending powers
bool polyfit_rows(const double *y, int64_t n, int polyorder, double *coef);
bool plan_rational_opair(const TypeKey &tk, bool fused, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds, int force_wgpc = 0);
bool plan_rational_opair_blocks(const TypeKey &tk, bool fused, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds, int *nblocks);   // L > 512: a workgroup owns a block of the period
hipError_t launch_rational_opair(bool fused, const PolyArgs &a, const PairArgs &pa, dim3 block, size_t lds, hipStream_t s,
                                 const char **kname, int num_cus, unsigned *counters);   // FIRRational and FIRInterpolator, two outputs per lane; also performs shiftin!
// rational_opair_kernel: STRICT (the product: bit-identical to the reference) is instantiated for every tapsPerPhi of every M/L class;
// the opt-in FUSED numerics for M/L < 2 and tapsPerPhi a multiple of 4 only (other FUSED shapes run on poly_phase_stationary_kernel /
// poly_tiled_kernel): the FUSED half of the matrix was 18 MB of code objects.
constexpr bool opair_instantiated(bool fused, int smin, int T) { return !fused || (smin <= 1 && T % 4 == 0); }
bool plan_fir_stream(const TypeKey &tk, const PolyArgs &a, int num_cus, PairArgs *out, dim3 *block, size_t *lds);
hipError_t launch_fir_stream(bool fused, const PolyArgs &a, const PairArgs &pa, dim3 block, size_t lds, hipStream_t s,
                             const char **kname, int num_cus, unsigned *counters);   // FIRStandard / FIRDecimator, streaming form; also performs shiftin!
bool plan_arb_tiled(const TypeKey &tk, const ArbArgs &a, const int32_t *n_idx_host, const int *spans, int num_cus, ArbTileArgs *out, size_t *lds);
bool plan_arb_pipe(const TypeKey &tk, const ArbArgs &a, long long span256, ArbTileArgs *out, size_t *lds);
hipError_t launch_arb_pipe(const TypeKey &tk, bool fused, const ArbArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                           const char **kname, int num_cus);
// arb_lane_kernel (kernels_arb_lane.hip): FIRArbitrary, Float64 x Float64, a lane per channel
struct ArbLaneArgs {
    int ring;                // samples per channel in the ring (a multiple of 16, >= the history length)
    int pitch8;              // 8-byte units between the channel rows of the ring: ring + mirror, odd
    int stretch;             // outputs per stretch (a multiple of 16)
    int chunk_max;           // most consecutive stretches handed out at once (the ring carries on inside a chunk)
    int ngroups;             // groups of 64 channels
    int y16;                 // 16-byte stores into y are aligned
    unsigned *counters;      // [0] stretches handed out beyond the first grid-full, [64] workgroups through (zero between launches)
};
bool plan_arb_lane(const TypeKey &tk, const ArbArgs &a, double rate, ArbLaneArgs *out, size_t *lds);
hipError_t launch_arb_lane(bool fused, const ArbArgs &a, const ArbLaneArgs &la, size_t lds, hipStream_t s, const char **kname, int num_cus);
hipError_t launch_arb_tiled(const TypeKey &tk, bool fused, const ArbArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                            const char **kname, int num_cus);
bool plan_poly_tiled(const TypeKey &tk, const PolyArgs &a, int num_cus, ArbTileArgs *out, size_t *lds);
hipError_t launch_poly_tiled(const TypeKey &tk, bool fused, const PolyArgs &a, const ArbTileArgs &ta, size_t lds, hipStream_t s,
                             const char **kname, int num_cus);
bool plan_phase_stationary(const TypeKey &tk, const PolyArgs &a, int num_cus, TileArgs *out, dim3 *grid, dim3 *block,
                           size_t *lds);
hipError_t launch_poly_phase_stationary(const TypeKey &tk, bool fused, const PolyArgs &a, const TileArgs &ta, dim3 grid,
                                        dim3 block, size_t lds, hipStream_t s, const char **kname, int num_cus);

}  // namespace mrhip
