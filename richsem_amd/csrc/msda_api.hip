// msda_api.hip -- the C ABI of librichsem_msda.so (declared in include/richsem_msda.h).
//
// Host side of the drop-in for the reference's ms_deform_attn_cuda_forward / _backward
// (reference models/richsem/ops/src/cuda/ms_deform_attn_cuda.cu:20-80, 83-153): argument checks,
// launch geometry, zero-fill of the accumulated output, kernel selection.  No torch types.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/richsem_msda.h"
#include "msda_direct.h"
#include "msda_levelsum.h"
#include "msda_prep.h"
#include "msda_dn.h"
#include "msda_topk.h"
#include "msda_roi.h"
#include "msda_band.h"
#include "msda_rps.h"
#include "msda_tiled.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace

// Error note of the library's other translation units (conv / ffn / lin256 / cls / attn / rows): "<entry point>: <class of error>", so
// that msda_last_error() after a failed call names the call and is never a stale message of an earlier one.  Not part of the C ABI.
extern "C" __attribute__((visibility("hidden"))) int msda_note_error(int code, const char *entry)
{
    const char *what = code == MSDA_ERR_NULL_POINTER ? "null pointer argument"
                     : code == MSDA_ERR_BAD_DIMS     ? "dimension out of the supported range (see include/richsem_msda.h)"
                     : code == MSDA_ERR_MISALIGNED   ? "pointer not aligned as required"
                     : code == MSDA_ERR_TOO_LARGE    ? "problem too large for 32-bit indexing"
                                                     : "error";
    return fail(code, "%s: %s", entry, what);
}

namespace {

int hip_fail(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
    return (int)e;
}

std::atomic<int> g_fwd_variant{0}, g_bwd_variant{0}, g_bwd_cpl{0}, g_levelsum{1}, g_fwd_prep_fused{1}, g_bwd_split{1};
thread_local int g_tl_fwd_variant = -1;      // >= 0: the forward variant of THIS call (set by a caller inside the library that has already chosen)

// ---- launch profiler: pre-created event pairs, one per logged call -----------------------------
struct ProfileSlot {
    hipEvent_t start, stop;
    msda_profile_record rec;
};
std::mutex g_prof_mutex;
std::vector<ProfileSlot> g_prof_slots;
int g_prof_used = 0;
std::atomic<bool> g_prof_on{false};
std::atomic<int> g_prof_filter{0};      // 0: every call is bracketed; else only calls of (kind + 1) * 16 + variant (option "profile_filter")

// RAII bracket around the main kernel launch of one call
struct ProfileScope {
    ProfileSlot *slot = nullptr;
    hipStream_t stream;
    ProfileScope(int kind, int variant, int dtype_bytes, int N, int S, int M, int D, int L, int Lq, int P,
                 hipStream_t st)
        : stream(st)
    {
        if (!g_prof_on.load(std::memory_order_relaxed)) return;
        // (an event pair is two more packets on the stream -- ~4 us of a call's time on MI355X: a caller that times a whole step brackets
        // only the kernel it wants the duration of)
        const int f = g_prof_filter.load(std::memory_order_relaxed);
        if (f && f != (kind + 1) * 16 + variant) return;
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        if (g_prof_used >= (int)g_prof_slots.size()) return;
        slot = &g_prof_slots[g_prof_used++];
        slot->rec = msda_profile_record{kind, variant, dtype_bytes, N, S, M, D, L, Lq, P, 0.f};
        (void)hipEventRecord(slot->start, stream);
    }
    // the call turned out not to run this variant: give the slot back (only the newest slot can be returned)
    void cancel()
    {
        if (!slot) return;
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        if (g_prof_used > 0 && slot == &g_prof_slots[g_prof_used - 1]) --g_prof_used;
        else slot->rec.variant = -1;
        slot = nullptr;
    }
    ~ProfileScope()
    {
        if (slot) (void)hipEventRecord(slot->stop, stream);
    }
};


// ---- locality monitor -----------------------------------------------------------------------------------
// The LDS-window FORWARD kernel wins only while most sampling points fall into the window of their query's region; the share
// that does not ("general share") is a property of the DATA (how far the network's offsets reach).  Measured on MI355X
// (tools/locality_sweep2.sh, call E; profiles/r02_locality.md): the window forward -- which finishes the points that miss
// their window per point, after the item -- beats the direct forward up to a share of ~8 % (sigma ~4.5 px).  In automatic mode
// the library therefore lets the window forward kernel count its general points now and then (one atomic per wave), brings the
// count back with an asynchronous copy + event on the caller's stream, and reads it on a LATER call once the event has completed
// -- no call ever waits.  Nothing is probed while the stream is being captured into a graph.  (The backward needs no monitor: the
// routed kernels of msda_rps.h cost the same wherever the points fall.)
constexpr float kFwdShareMax = 0.08f;
constexpr unsigned kProbeWarmCalls = 2, kProbeEvery = 64;   // per (shape, sampling_loc buffer)
constexpr int kMaxDevices = 64;
std::atomic<int> g_monitor_on{1};
std::atomic<int> g_last_share_ppm{-1};

struct MonitorEntry {
    unsigned calls = 0, next_probe = 0, probes = 0;
    float share = 0.f;
    bool known = false;
};
constexpr int kProbeSlots = 8;   // probes in flight at once (the layers of a step call back to back)
struct ProbeSlot {
    hipEvent_t done = nullptr;
    bool pending = false;
    uint64_t key = 0;
    double points = 0;
};
struct Monitor {
    std::mutex mu;
    unsigned *dev = nullptr, *host = nullptr;   // kProbeSlots counters on the device, their pinned host copies
    ProbeSlot slot[kProbeSlots];
    int held = -1;                               // slot of the probe being launched (between choose_fwd and finish_probe)
    std::unordered_map<uint64_t, MonitorEntry> table;
};
Monitor g_monitors[kMaxDevices];

// What a verdict is keyed by: the problem's shape AND the sampling_loc buffer.  How far the offsets reach is a property of a
// LAYER's weights; the layers of a network call with the same shapes but each with its own sampling_loc tensor, which a
// steady-state training step finds at the same address every time (caching allocator) -- and the backward call of a layer
// is handed the very tensor its forward call saw.  A pointer never seen before simply starts a new entry.
uint64_t problem_key(const int N, const int S, const int M, const int L, const int P, const int64_t *shapes, const void *loc = nullptr)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix(N); mix(S); mix(M); mix(L); mix(P);
    for (int l = 0; l < 2 * L; ++l) mix((uint64_t)shapes[l]);
    mix((uint64_t)reinterpret_cast<uintptr_t>(loc));
    return h;
}
constexpr size_t kMonitorMaxEntries = 512;   // (sampling_loc addresses that keep changing: forget and start over)

// under mo.mu: fold the finished probes into their entries
void monitor_poll(Monitor &mo)
{
    for (int i = 0; i < kProbeSlots; ++i) {
        ProbeSlot &ps = mo.slot[i];
        if (!ps.pending || hipEventQuery(ps.done) != hipSuccess) continue;
        ps.pending = false;
        MonitorEntry &en = mo.table[ps.key];
        const float s = (float)((double)mo.host[i] / ps.points);
        // the warm-up probes of a key keep their worst; later ones track slowly
        en.share = !en.known ? s : (en.probes <= kProbeWarmCalls ? (s > en.share ? s : en.share) : 0.5f * (en.share + s));
        en.known = true;
        g_last_share_ppm = (int)(s * 1e6f);
    }
}

Monitor *monitor_for_current_device()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    return &g_monitors[dev];
}

// Forward, automatic mode, window kernels applicable.  Returns the variant to run (1 direct, 2 window) and, through
// *probe, the device counter the window kernel should add its general points to (or null).  With *probe set the caller
// MUST call monitor_finish_probe after its launch.  The monitor's mutex is held from here to monitor_finish_probe.
int monitor_choose_fwd(Monitor *mo, uint64_t key, hipStream_t stream, unsigned **probe)
{
    *probe = nullptr;
    if (!mo || !g_monitor_on.load()) return 2;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    mo->mu.lock();
    if (!capturing) monitor_poll(*mo);   // (querying an event is not allowed while a capture is under way)
    if (mo->table.size() > kMonitorMaxEntries) mo->table.clear();
    MonitorEntry &en = mo->table[key];
    const unsigned call = en.calls++;
    int free_slot = -1;
    bool mine_in_flight = false;
    for (int i = 0; i < kProbeSlots; ++i) {
        if (!mo->slot[i].pending && free_slot < 0) free_slot = i;
        mine_in_flight = mine_in_flight || (mo->slot[i].pending && mo->slot[i].key == key);
    }
    bool want = free_slot >= 0 && !mine_in_flight && !capturing && call >= en.next_probe;
    if (want && !mo->dev) {   // first probe on this device
        bool ok = hipMalloc(reinterpret_cast<void **>(&mo->dev), kProbeSlots * sizeof(unsigned)) == hipSuccess &&
                  hipHostMalloc(reinterpret_cast<void **>(&mo->host), kProbeSlots * sizeof(unsigned), hipHostMallocDefault) == hipSuccess;
        for (int i = 0; ok && i < kProbeSlots; ++i) ok = hipEventCreateWithFlags(&mo->slot[i].done, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            mo->dev = nullptr;
            want = false;
        }
    }
    if (want && hipMemsetAsync(mo->dev + free_slot, 0, sizeof(unsigned), stream) != hipSuccess) want = false;
    if (want) {
        *probe = mo->dev + free_slot;
        en.next_probe = call + (en.probes < kProbeWarmCalls ? 1 : kProbeEvery);
        ++en.probes;
        mo->slot[free_slot].key = key;
        mo->held = free_slot;
        return 2;   // a probe IS a window-kernel call; the lock stays held
    }
    const int variant = en.known && en.share > kFwdShareMax ? 1 : 2;
    mo->mu.unlock();
    return variant;
}

void monitor_finish_probe(Monitor *mo, double points, hipStream_t stream, bool launched)
{
    const int i = mo->held;
    mo->held = -1;
    if (i >= 0 && launched &&
        hipMemcpyAsync(mo->host + i, mo->dev + i, sizeof(unsigned), hipMemcpyDeviceToHost, stream) == hipSuccess &&
        hipEventRecord(mo->slot[i].done, stream) == hipSuccess) {
        mo->slot[i].pending = true;
        mo->slot[i].points = points;
    }
    mo->mu.unlock();
}

struct Problem {
    int N, S, M, D, L, Lq, P;
    std::vector<int64_t> shapes, lsi;  // host mirrors
};

bool is_aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Argument checks shared by forward and backward.  Mirrors the reference's preconditions
// (ms_deform_attn_cuda.cu:28-52) and adds the bounds the kernels rely on.
int check_problem(Problem &pb, const int64_t *shapes_dev, const int64_t *lsi_dev, const int64_t *shapes_host,
                  const int64_t *lsi_host, int im2col_step, hipStream_t stream)
{
    if (pb.N <= 0 || pb.S <= 0 || pb.M <= 0 || pb.D <= 0 || pb.L <= 0 || pb.Lq <= 0 || pb.P <= 0)
        return fail(MSDA_ERR_BAD_DIMS, "non-positive dimension (N=%d S=%d M=%d D=%d L=%d Lq=%d P=%d)", pb.N, pb.S,
                    pb.M, pb.D, pb.L, pb.Lq, pb.P);
    if (im2col_step <= 0) return fail(MSDA_ERR_IM2COL_STEP, "im2col_step(%d) must be positive", im2col_step);
    const int step = pb.N < im2col_step ? pb.N : im2col_step;
    if (pb.N % step != 0) return fail(MSDA_ERR_IM2COL_STEP, "batch(%d) must divide im2col_step(%d)", pb.N, step);

    const int64_t lim = (int64_t)1 << 31;
    const int64_t n_value = (int64_t)pb.N * pb.S * pb.M * pb.D;
    const int64_t n_out = (int64_t)pb.N * pb.Lq * pb.M * pb.D;
    const int64_t n_loc = (int64_t)pb.N * pb.Lq * pb.M * pb.L * pb.P * 2;
    if (n_value >= lim || n_out >= lim || n_loc >= lim)
        return fail(MSDA_ERR_TOO_LARGE, "tensor with >= 2^31 elements (value %lld, out %lld, loc %lld)",
                    (long long)n_value, (long long)n_out, (long long)n_loc);

    pb.shapes.resize(2 * (size_t)pb.L);
    pb.lsi.resize((size_t)pb.L);
    if (shapes_host && lsi_host) {
        memcpy(pb.shapes.data(), shapes_host, sizeof(int64_t) * 2 * pb.L);
        memcpy(pb.lsi.data(), lsi_host, sizeof(int64_t) * pb.L);
    } else {  // slow path: synchronises the stream
        hipError_t e = hipMemcpyAsync(pb.shapes.data(), shapes_dev, sizeof(int64_t) * 2 * pb.L, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(pb.lsi.data(), lsi_dev, sizeof(int64_t) * pb.L, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return hip_fail(e, "copying spatial_shapes / level_start_index to the host");
    }
    int64_t total = 0;
    for (int l = 0; l < pb.L; ++l) {
        const int64_t H = pb.shapes[2 * l], W = pb.shapes[2 * l + 1], st = pb.lsi[l];
        if (H <= 0 || W <= 0 || H >= lim || W >= lim || H * W > pb.S)
            return fail(MSDA_ERR_BAD_DIMS, "level %d: bad spatial shape (%lld, %lld)", l, (long long)H, (long long)W);
        if (st < 0 || st + H * W > pb.S)
            return fail(MSDA_ERR_BAD_DIMS, "level %d: level_start_index %lld + %lld*%lld exceeds S=%d", l, (long long)st,
                        (long long)H, (long long)W, pb.S);
        total += H * W;
    }
    if (total != pb.S)
        return fail(MSDA_ERR_BAD_DIMS, "sum of H*W over levels (%lld) != S (%d)", (long long)total, pb.S);
    return MSDA_OK;
}

int floor_log2(int x)
{
    int r = 0;
    while ((1 << (r + 1)) <= x) ++r;
    return r;
}

// Channels per lane for the direct kernels: the widest access (<= 16 B) that D and every data
// pointer allow.
template <typename T>
int pick_channels_per_lane(int D, std::initializer_list<const void *> ptrs)
{
    int c = (int)(16 / sizeof(T));
    for (; c > 1; c >>= 1) {
        bool ok = D % c == 0;
        for (const void *p : ptrs) ok = ok && is_aligned(p, sizeof(T) * c);
        if (ok) break;
    }
    return c;
}

msda::DirectGeom direct_geom(const Problem &pb, int C)
{
    msda::DirectGeom g{};
    g.N = pb.N; g.S = pb.S; g.M = pb.M; g.D = pb.D; g.L = pb.L; g.Lq = pb.Lq; g.P = pb.P;
    const int lanes = (pb.D + C - 1) / C;
    int G = msda::kMinGroup;
    while (G < lanes && G < msda::kWave) G <<= 1;
    g.G = G;
    g.logG = floor_log2(G);
    g.nchunks = (lanes + G - 1) / G;
    // queries per block: whole passes of the block's lane groups; few passes when Lq is small (decoder calls) so that
    // the grid still fills 256 CUs several times over, more passes (amortising the level table) when Lq is large
    const int per_iter = msda::kDirectWaves * (msda::kWave / G);
    const int slots = msda::kXcds * ((pb.N * pb.M + msda::kXcds - 1) / msda::kXcds);
    const int want_tiles = (4096 + slots - 1) / slots;
    int passes = pb.Lq / (per_iter * want_tiles);
    passes = passes < 1 ? 1 : (passes > 8 ? 8 : passes);
    g.qtile = per_iter * passes;
    g.ntiles = (pb.Lq + g.qtile - 1) / g.qtile;
    const int LP = pb.L * pb.P;
    g.pbatch = LP < msda::kPointBatch ? LP : msda::kPointBatch;
    return g;
}

// fp32 only: whole-level LDS sums of grad_value (msda_levelsum.h); `taken` = bit mask of the levels it produced
template <typename T>
hipError_t launch_levelsum(const Problem &, const T *, const T *, const T *, T *, hipStream_t, unsigned &taken)
{
    taken = 0;
    return hipSuccess;
}
template <>
hipError_t launch_levelsum<float>(const Problem &pb, const float *loc, const float *aw, const float *grad_out,
                                  float *grad_value, hipStream_t stream, unsigned &taken)
{
    msda::LevelSumGeom lg;
    size_t lds = 0;
    taken = msda::plan_levelsum(pb.N, pb.S, pb.M, pb.D, pb.L, pb.Lq, pb.P, pb.shapes.data(), pb.lsi.data(), lg, lds);
    if (!taken) return hipSuccess;
    // the P4 form loads a level's four locations / weights as 16-B vectors: only for 16-B aligned tensors (the ABI asks
    // for element alignment only)
    const bool vec = pb.P == 4 && is_aligned(loc, 16) && is_aligned(aw, 16);
    lg.dbg = msda::tiled_options().dbg & 7;
    auto kern = vec ? &msda::bwd_levelsum_kernel<true> : &msda::bwd_levelsum_kernel<false>;
    hipError_t e = msda::set_lds_limit(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(msda::levelsum_grid(lg)), dim3(msda::kLsThreads), lds, stream, loc, aw, grad_out, grad_value,
                       lg);
    return hipGetLastError();
}

// the levels launch_levelsum would take for this problem (host-only)
template <typename T>
unsigned levelsum_levels(const Problem &)
{
    return 0;
}
template <>
unsigned levelsum_levels<float>(const Problem &pb)
{
    msda::LevelSumGeom lg;
    size_t lds = 0;
    return msda::plan_levelsum(pb.N, pb.S, pb.M, pb.D, pb.L, pb.Lq, pb.P, pb.shapes.data(), pb.lsi.data(), lg, lds);
}

int direct_grid(const msda::DirectGeom &g)
{
    const int pairs = g.N * g.M;
    return msda::kXcds * ((pairs + msda::kXcds - 1) / msda::kXcds) * g.ntiles;
}

// ---- library-owned device memory, per (device, stream): calls on one stream are ordered, so they can share it; calls on
// different streams cannot.  Nothing is allocated while the stream is being captured.
struct Workspace {
    float *gv32 = nullptr;   // bf16 backward: fp32 accumulation buffer for grad_value (rounded to bf16 once)
    size_t gv32_cap = 0;
    unsigned *rps_bins = nullptr;            // routed backward: queue heads (32 words), then one 64-bit (records | runs) counter per
    size_t rps_bins_cap = 0;                 // bin on a 128-byte line of its own
    size_t rps_bins_n = 0;                   // capacity in bins
    uint2 *rps_runs = nullptr;               // [cap] the bins' run tables (nbins x max_runs)
    size_t rps_runs_cap = 0;
    msda::RpsRec *rps_entries = nullptr;     // [cap] routed records
    size_t rps_entries_cap = 0;
    bool rps_dirty = false;                  // a failed launch may have left the bin counters non-zero
    // Buffers outgrown by a later, larger call are RETIRED, not freed: a HIP graph captured earlier on this stream holds their raw
    // addresses (kernel arguments), and a replay after a hipFree would read and write freed memory.  They live until the process ends;
    // growth is geometric (x 1.5), so the retired total stays below twice the live size.
    std::vector<void *> retired;
    void retire(void *p) { if (p) retired.push_back(p); }
};
std::mutex g_ws_mu;
std::map<std::pair<int, hipStream_t>, Workspace> g_ws;
int g_cu_count[kMaxDevices] = {0};

int cu_count()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    if (!g_cu_count[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cu_count[dev] = n;
    }
    return g_cu_count[dev];
}

// ---- routed pixel-stationary backward (msda_rps.h) ---------------------------------------------------------------------------
// The bin counters are zero between calls (the tile kernel zeroes a bin's counter when it has consumed the bin), so that no call has to clear
// them first: they are cleared when allocated and after a launch error.
bool rps_workspace(hipStream_t stream, size_t n_bins, size_t n_runs, size_t n_entries, Workspace &out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_ws_mu);
    Workspace &ws = g_ws[std::make_pair(dev, stream)];
    if (ws.rps_bins_n < n_bins || ws.rps_runs_cap < n_runs || ws.rps_entries_cap < n_entries || ws.rps_dirty) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cap);
        if (cap != hipStreamCaptureStatusNone) return false;   // no allocation while the stream is being captured
        if (ws.rps_bins_n < n_bins) {
            const size_t cap_n = n_bins + n_bins / 2 + 64;
            const size_t want_bins = 32 + (size_t)msda::kRpsPad * cap_n + 8;
            ws.retire(ws.rps_bins);
            ws.rps_bins = nullptr;
            ws.rps_bins_cap = ws.rps_bins_n = 0;
            if (hipMalloc(reinterpret_cast<void **>(&ws.rps_bins), want_bins * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return false; }
            ws.rps_bins_cap = want_bins;
            ws.rps_bins_n = cap_n;
            ws.rps_dirty = true;
        }
        if (ws.rps_dirty) {
            if (hipMemsetAsync(ws.rps_bins, 0, ws.rps_bins_cap * sizeof(unsigned), stream) != hipSuccess) { (void)hipGetLastError(); return false; }
            ws.rps_dirty = false;
        }
        if (ws.rps_runs_cap < n_runs) {
            ws.retire(ws.rps_runs);
            ws.rps_runs = nullptr;
            ws.rps_runs_cap = 0;
            const size_t want = n_runs + n_runs / 2 + 1024;
            if (hipMalloc(reinterpret_cast<void **>(&ws.rps_runs), want * sizeof(uint2)) != hipSuccess) { (void)hipGetLastError(); return false; }
            ws.rps_runs_cap = want;
        }
        if (ws.rps_entries_cap < n_entries) {   // (+ the tile kernel's scratch lines behind the records)
            ws.retire(ws.rps_entries);
            ws.rps_entries = nullptr;
            ws.rps_entries_cap = 0;
            const size_t want = n_entries + n_entries / 2;
            if (hipMalloc(reinterpret_cast<void **>(&ws.rps_entries), want * sizeof(msda::RpsRec) + msda::kRpsDummyBytes) != hipSuccess) { (void)hipGetLastError(); return false; }
            ws.rps_entries_cap = want;
        }
    }
    out = ws;
    return true;
}

void rps_mark_dirty(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_ws_mu);
    g_ws[std::make_pair(dev, stream)].rps_dirty = true;
}

// returns hipErrorNotSupported when the plan does not apply or no workspace can be had right now.
// TV = bf16_t: grad_acc is an fp32 image of grad_value for the levels accumulated with atomics (rounded at the end).
template <typename TV>
hipError_t launch_bwd_rps(const Problem &pb, const TV *value, const float *loc, const float *aw, const TV *grad_out,
                          TV *grad_value, float *grad_acc, float *grad_loc, float *grad_aw, hipStream_t stream)
{
    msda::RpsPlan pl = msda::plan_rps(pb.N, pb.S, pb.M, pb.D, pb.L, pb.Lq, pb.P, pb.shapes.data(), pb.lsi.data());
    if (!pl.ok) return hipErrorNotSupported;
    constexpr uintptr_t row_align = sizeof(TV) * 4 - 1;   // 16 B (fp32) / 8 B (bf16) per lane access
    if ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_value)) & row_align)
        return hipErrorNotSupported;
    if (reinterpret_cast<uintptr_t>(grad_acc) & 15) return hipErrorNotSupported;
    if ((reinterpret_cast<uintptr_t>(grad_loc) | reinterpret_cast<uintptr_t>(loc)) & 7) return hipErrorNotSupported;
    if ((int64_t)pb.N * pb.Lq * pb.M * msda::kRpsD * (int64_t)sizeof(TV) > (int64_t)0xFFFFFFFF) return hipErrorNotSupported;      // (the tile kernel addresses grad_out rows by 32-bit byte offsets)
    Workspace ws;
    const size_t n_runs = (size_t)pl.g.nbins * (size_t)pl.g.max_runs;
    if (n_runs * sizeof(uint2) > ((size_t)256 << 20)) return hipErrorNotSupported;      // (run tables of a quarter GB: not a shape this path is for)
    if (!rps_workspace(stream, (size_t)pl.g.nbins, n_runs, pl.max_entries, ws)) return hipErrorNotSupported;
    pl.g.ctr = ws.rps_bins;
    pl.g.bin_state = reinterpret_cast<unsigned long long *>(ws.rps_bins + 32);   // (line-aligned; one per 128-B line)
    pl.g.runs = ws.rps_runs;
    pl.g.entries_cap = (unsigned)std::min<size_t>(ws.rps_entries_cap, 0xFFFFFFFFu);
    pl.g.entries = ws.rps_entries;
    pl.g.dummy = reinterpret_cast<float *>(ws.rps_entries + ws.rps_entries_cap);
    pl.g.stamps = msda::tiled_options().stamps;
    pl.g.dbg = msda::tiled_options().dbg;
    auto kern = pl.g.stamps ? (pb.P == 4 ? &msda::rps_tile_kernel<true, TV, true> : &msda::rps_tile_kernel<false, TV, true>)
                            : (pb.P == 4 ? &msda::rps_tile_kernel<true, TV> : &msda::rps_tile_kernel<false, TV>);
    hipError_t e = msda::set_lds_limit(reinterpret_cast<const void *>(kern), sizeof(msda::RpsLds));
    if (e != hipSuccess) return e;
    // route passes: a workgroup (8 waves; 16 where the plan asks for them) per block of queries of an (image, head)
    const int qpw = pb.P <= 4 ? 16 : (pb.P <= 8 ? 8 : (pb.P <= 16 ? 4 : (pb.P <= 32 ? 2 : 1)));
    const int qpb = qpw * (pl.g.route_threads / msda::kWave);
    const int64_t r_items = (int64_t)pb.N * pb.M * ((pb.Lq + qpb - 1) / qpb);
    const int route_wgs = pl.g.route_threads > 512 ? 1 : msda::rps_options().route_wgs.load();      // (16 waves at 102 registers: one workgroup per CU)
    const int rgrid = (int)std::max<int64_t>(1, std::min<int64_t>(r_items, (int64_t)route_wgs * cu_count()));
    {
        // (the reference's models all have L = P = 4: that instance has both as compile-time constants; the stage stamps live in an
        // instance of their own)
        using RouteFn = void (*)(const float *, const float *, float *, float *, float *, const msda::RpsGeom);
        const bool big = pl.g.route_threads > 512, lp4 = pb.L == 4 && pb.P == 4;
        RouteFn route;
        if (pl.g.stamps)
            route = big ? &msda::rps_route_kernel<msda::kRpsRouteThreadsMax, 0, 0, true>
                        : (lp4 ? &msda::rps_route_kernel<512, 4, 4, true> : &msda::rps_route_kernel<512, 0, 0, true>);
        else if (lp4)
            route = big ? &msda::rps_route_kernel<msda::kRpsRouteThreadsMax, 4, 4> : &msda::rps_route_kernel<512, 4, 4>;
        else
            route = big ? &msda::rps_route_kernel<msda::kRpsRouteThreadsMax> : &msda::rps_route_kernel<512>;
        hipLaunchKernelGGL(route, dim3(rgrid), dim3(pl.g.route_threads), (size_t)pl.g.lut_n * sizeof(unsigned), stream, loc, aw, grad_acc,
                           grad_loc, grad_aw, pl.g);
    }
    const int grid = (cu_count() / msda::kXcds) * msda::kXcds;   // persistent: one workgroup per CU (its LDS is most of a CU's)
#ifdef RPS_ROUTE_ABLATION
    if (pl.g.dbg & 0x300) {      // diagnostic: route-pass ablations -- the records are not what the tile kernel expects; wrong results
        rps_mark_dirty(stream);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid > 0 ? grid : 8), dim3(msda::kRpsThreads), sizeof(msda::RpsLds), stream, value, grad_out,
                       grad_value, grad_acc, grad_loc, grad_aw, pl.g);
    if constexpr (!std::is_same<TV, float>::value)      // (every level has rows in the fp32 image: the tiles' shared first rows / columns at least)
        hipLaunchKernelGGL(msda::rps_round_kernel, dim3(256), dim3(256), 0, stream, grad_acc, grad_value, pl.g);
    e = hipGetLastError();
    if (e != hipSuccess) rps_mark_dirty(stream);
    return e;
}

// ---- row-band backward (msda_band.h): decoder-shaped calls, D = 32, one launch ------------------------------------------------------
// returns hipErrorNotSupported when the plan does not apply.  TV = bf16_t: grad_acc is an fp32 image of grad_value for the slabbed levels.
template <typename TV>
hipError_t launch_bwd_band(const Problem &pb, const TV *value, const float *loc, const float *aw, const TV *grad_out, TV *grad_value,
                           float *grad_acc, float *grad_loc, float *grad_aw, hipStream_t stream)
{
    msda::BandPlan pl = msda::plan_band(pb.N, pb.S, pb.M, pb.D, pb.L, pb.Lq, pb.P, pb.shapes.data(), pb.lsi.data());
    if (!pl.ok) return hipErrorNotSupported;
    pl.g.dbg = msda::tiled_options().dbg & 63;
    pl.g.stamps = msda::tiled_options().stamps;
    if ((reinterpret_cast<uintptr_t>(grad_loc) | reinterpret_cast<uintptr_t>(loc)) & 7) return hipErrorNotSupported;
    constexpr uintptr_t row_align = sizeof(TV) * 4 - 1;   // 16 B (fp32) / 8 B (bf16) per lane access
    if ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_value)) & row_align)
        return hipErrorNotSupported;
    if (pl.atomic_levels && !grad_acc) return hipErrorNotSupported;
    if (reinterpret_cast<uintptr_t>(grad_acc) & 15) return hipErrorNotSupported;
    hipError_t e;
    if (pl.atomic_levels)      // the levels several workgroups add to: zero in every image
        hipLaunchKernelGGL(msda::band_zero_kernel, dim3(128), dim3(256), 0, stream, grad_acc, pl.g, pl.atomic_levels);
    auto kern = &msda::bwd_band_kernel<TV>;
    e = msda::set_lds_limit(reinterpret_cast<const void *>(kern), pl.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(msda::band_grid(pl.g)), dim3(msda::kBandThreads), pl.lds_bytes, stream, value, loc, aw, grad_out, grad_value,
                       grad_acc, grad_loc, grad_aw, pl.g);
    if constexpr (!std::is_same<TV, float>::value) {
        if (pl.atomic_levels) hipLaunchKernelGGL(msda::band_round_kernel, dim3(256), dim3(256), 0, stream, grad_acc, grad_value, pl.g, pl.atomic_levels);
    }
    return hipGetLastError();
}

template <typename T>
hipError_t try_bwd_band(const Problem &, const T *, const T *, const T *, const T *, T *, T *, T *, hipStream_t)
{
    return hipErrorNotSupported;
}
template <>
hipError_t try_bwd_band<float>(const Problem &pb, const float *value, const float *loc, const float *aw, const float *grad_out,
                               float *grad_value, float *grad_loc, float *grad_aw, hipStream_t stream)
{
    return launch_bwd_band<float>(pb, value, loc, aw, grad_out, grad_value, grad_value, grad_loc, grad_aw, stream);
}

template <typename T>
hipError_t try_bwd_rps(const Problem &, const T *, const T *, const T *, const T *, T *, T *, T *, hipStream_t)
{
    return hipErrorNotSupported;
}
template <>
hipError_t try_bwd_rps<float>(const Problem &pb, const float *value, const float *loc, const float *aw, const float *grad_out,
                              float *grad_value, float *grad_loc, float *grad_aw, hipStream_t stream)
{
    return launch_bwd_rps<float>(pb, value, loc, aw, grad_out, grad_value, grad_value, grad_loc, grad_aw, stream);
}

// The split kernels (msda_direct.h: fwd_split_kernel / bwd_split_kernel) apply at D = 32 with 16-byte (fp32) / 8-byte (bf16) rows, L*P a
// multiple of 4 up to 32; they are the automatic choice for calls of fewer than 65536 (query, head) items.
inline bool split_fits(const Problem &pb, int C) { return pb.D == 32 && C == 4 && (pb.L * pb.P) % msda::kSplitGroups == 0 && pb.L * pb.P / msda::kSplitGroups <= msda::kSplitMaxPts; }
inline bool split_small(const Problem &pb) { return (int64_t)pb.N * pb.Lq * pb.M < 65536; }
inline dim3 split_grid(const Problem &pb) { return dim3((unsigned)(((int64_t)pb.N * pb.Lq * pb.M * 32 + msda::kDirectThreads - 1) / msda::kDirectThreads)); }

template <typename TV>
hipError_t launch_fwd_split(const Problem &pb, const TV *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *aw,
                            TV *out, const msda::DirectGeom &g, hipStream_t stream)
{
    const size_t lds = sizeof(msda::LevelGeom) * pb.L;
    if (pb.L * pb.P == 16)
        hipLaunchKernelGGL((msda::fwd_split_kernel<TV, 4>), split_grid(pb), dim3(msda::kDirectThreads), lds, stream, value, shapes, lsi, loc, aw, out, g);
    else
        hipLaunchKernelGGL((msda::fwd_split_kernel<TV, 0>), split_grid(pb), dim3(msda::kDirectThreads), lds, stream, value, shapes, lsi, loc, aw, out, g);
    return hipGetLastError();
}

template <typename TV>
hipError_t launch_bwd_split(const Problem &pb, const TV *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *aw,
                            const TV *grad_out, float *grad_loc, float *grad_aw, const msda::DirectGeom &g, hipStream_t stream)
{
    const size_t lds = sizeof(msda::LevelGeom) * pb.L;
    if (pb.L * pb.P == 16)
        hipLaunchKernelGGL((msda::bwd_split_kernel<TV, 4>), split_grid(pb), dim3(msda::kDirectThreads), lds, stream, value, shapes, lsi, loc, aw,
                           grad_out, grad_loc, grad_aw, g);
    else
        hipLaunchKernelGGL((msda::bwd_split_kernel<TV, 0>), split_grid(pb), dim3(msda::kDirectThreads), lds, stream, value, shapes, lsi, loc, aw,
                           grad_out, grad_loc, grad_aw, g);
    return hipGetLastError();
}

template <typename T>
int forward_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *aw, int N, int S,
                 int M, int D, int L, int Lq, int P, int im2col_step, T *out, const int64_t *shapes_host,
                 const int64_t *lsi_host, msda_stream_t stream_)
{
    g_err[0] = 0;
    if (!value || !shapes || !lsi || !loc || !aw || !out) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Problem pb{N, S, M, D, L, Lq, P, {}, {}};
    if (int rc = check_problem(pb, shapes, lsi, shapes_host, lsi_host, im2col_step, stream)) return rc;
    if (!is_aligned(value, sizeof(T)) || !is_aligned(out, sizeof(T)) || !is_aligned(aw, sizeof(T)) ||
        !is_aligned(loc, 2 * sizeof(T)) || !is_aligned(shapes, 8) || !is_aligned(lsi, 8))
        return fail(MSDA_ERR_MISALIGNED, "misaligned pointer (sampling_loc needs 2*sizeof(T))");

    int variant = g_tl_fwd_variant >= 0 ? g_tl_fwd_variant : g_fwd_variant.load();
    if (variant != 1 && msda::tiled_fwd_applicable<T>(pb.N, pb.S, pb.M, pb.D, pb.L, pb.Lq, pb.P, pb.shapes.data(),
                                                     pb.lsi.data(), value, out)) {
        unsigned *probe = nullptr;
        Monitor *mo = nullptr;
        if (variant == 0) {   // automatic: follow the locality monitor
            mo = monitor_for_current_device();
            variant = monitor_choose_fwd(mo, problem_key(N, S, M, L, P, pb.shapes.data(), loc), stream, &probe);
        }
        if (variant == 2) {
            hipError_t e;
            {
                ProfileScope prof(0, 2, (int)sizeof(T), N, S, M, D, L, Lq, P, stream);
                e = msda::launch_fwd_tiled<T>(value, shapes, lsi, loc, aw, out, pb.N, pb.S, pb.M, pb.D, pb.L, pb.Lq, pb.P,
                                              pb.shapes.data(), pb.lsi.data(), probe, stream);
            }
            if (probe) monitor_finish_probe(mo, 2.0 * N * Lq * M * L * P, stream, e == hipSuccess);
            if (e != hipSuccess) return hip_fail(e, "launch of the tiled forward kernel");
            return MSDA_OK;
        }
    }

    const int C = pick_channels_per_lane<T>(D, {value, out});
    msda::DirectGeom g = direct_geom(pb, C);
    g.head_major = (msda::tiled_options().dbg & 128) ? 1 : 0;      // (measured experiment: value read as (N, M, S, D))
    // small fp32 calls at D = 32 (decoder-shaped): 32 lanes per item, all of a lane's gathers in flight at once (fwd_split_kernel);
    // fwd_variant 3 forces it wherever it applies, 1 keeps the 8-lane kernel
    if constexpr (std::is_same<T, float>::value) {
        if (split_fits(pb, C) && !g.head_major && (variant == 3 || (variant != 1 && split_small(pb)))) {
            ProfileScope prof(0, 3, (int)sizeof(T), N, S, M, D, L, Lq, P, stream);
            const hipError_t e = launch_fwd_split<float>(pb, value, shapes, lsi, loc, aw, out, g, stream);
            if (e != hipSuccess) return hip_fail(e, "launch of the split forward kernel");
            return MSDA_OK;
        }
    }
    const size_t lds = msda::direct_lds_bytes<T>(g);
    if (lds > 64 * 1024) return fail(MSDA_ERR_BAD_DIMS, "too many levels (L=%d) for the level table in LDS", L);
    const dim3 grid(direct_grid(g)), block(msda::kDirectThreads);
    ProfileScope prof(0, 1, (int)sizeof(T), N, S, M, D, L, Lq, P, stream);
    // large calls: more waves per SIMD, shallower per-wave pipeline (see fwd_direct_kernel)
    const bool many = (int64_t)N * Lq * M >= 65536;
#define MSDA_LAUNCH_FWD(CC)                                                                                              \
    if (many) hipLaunchKernelGGL((msda::fwd_direct_kernel<T, CC, 6>), grid, block, lds, stream, value, shapes, lsi, loc, aw, out, g); \
    else hipLaunchKernelGGL((msda::fwd_direct_kernel<T, CC, 4>), grid, block, lds, stream, value, shapes, lsi, loc, aw, out, g)
    switch (C) {
        case 4: MSDA_LAUNCH_FWD((sizeof(T) == 4 ? 4 : 2)); break;
        case 2: MSDA_LAUNCH_FWD(2); break;
        default: MSDA_LAUNCH_FWD(1); break;
    }
#undef MSDA_LAUNCH_FWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the direct forward kernel");
    return MSDA_OK;
}

template <typename T>
int backward_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *aw,
                  const T *grad_out, int N, int S, int M, int D, int L, int Lq, int P, int im2col_step, T *grad_value,
                  T *grad_loc, T *grad_aw, const int64_t *shapes_host, const int64_t *lsi_host, msda_stream_t stream_)
{
    g_err[0] = 0;
    if (!value || !shapes || !lsi || !loc || !aw || !grad_out || !grad_value || !grad_loc || !grad_aw)
        return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Problem pb{N, S, M, D, L, Lq, P, {}, {}};
    if (int rc = check_problem(pb, shapes, lsi, shapes_host, lsi_host, im2col_step, stream)) return rc;
    if (!is_aligned(value, sizeof(T)) || !is_aligned(grad_out, sizeof(T)) || !is_aligned(aw, sizeof(T)) ||
        !is_aligned(grad_value, sizeof(T)) || !is_aligned(grad_aw, sizeof(T)) || !is_aligned(loc, 2 * sizeof(T)) ||
        !is_aligned(grad_loc, 2 * sizeof(T)) || !is_aligned(shapes, 8) || !is_aligned(lsi, 8))
        return fail(MSDA_ERR_MISALIGNED, "misaligned pointer (sampling_loc / grad_sampling_loc need 2*sizeof(T))");

    // grad_value is accumulated into (the reference gets it from at::zeros_like, ms_deform_attn_cuda.cu:121)
    hipError_t e = hipSuccess;
    auto zero_grad_value = [&]() { return hipMemsetAsync(grad_value, 0, sizeof(T) * (size_t)N * S * M * D, stream); };

    // Kernel choice: the routed pixel-stationary kernels (msda_rps.h: no float atomics, no zero-fill, cost independent of where
    // the points fall) for encoder-shaped fp32 calls at D = 32 (bwd_variant 4 forces them for any Lq, 1 forces the direct path);
    // everything else -- other D, f64, decoder shapes -- runs the level-sum + direct kernels below.
    const int variant = g_bwd_variant.load();
    if (variant == 4 || (variant == 0 && Lq == S)) {
        ProfileScope prof(1, 4, (int)sizeof(T), N, S, M, D, L, Lq, P, stream);
        e = try_bwd_rps<T>(pb, value, loc, aw, grad_out, grad_value, grad_loc, grad_aw, stream);
        if (e == hipSuccess) return MSDA_OK;
        prof.cancel();
        if (e != hipErrorNotSupported) return hip_fail(e, "launch of the routed backward kernels");
        e = hipSuccess;
    }

    // bwd_variant 5: the row-band kernel (msda_band.h) -- grad_value, grad_sampling_loc and grad_attn_weight in ONE launch, no zero-fill of
    // grad_value.  A measured option, not the automatic choice: 101-123 us on the decoder call Dd against 88 us for the level-sum + direct
    // kernels below (profiles/r05_dd_backward.md).
    if (variant == 5) {
        ProfileScope prof(1, 5, (int)sizeof(T), N, S, M, D, L, Lq, P, stream);
        e = try_bwd_band<T>(pb, value, loc, aw, grad_out, grad_value, grad_loc, grad_aw, stream);
        if (e == hipSuccess) return MSDA_OK;
        prof.cancel();
        if (e != hipErrorNotSupported) return hip_fail(e, "launch of the row-band backward kernel");
        e = hipSuccess;
    }

    // Levels summed in LDS by their own kernel (msda_levelsum.h) are taken away from the atomics below; when that is ALL
    // levels (decoder-shaped calls) grad_value is written, not accumulated: no zero-fill, no atomics.
    unsigned ls_levels = 0;
    const unsigned all_levels = L >= 32 ? ~0u : (1u << L) - 1;
    // (its plain per-level stores need the levels to tile [0, S) without overlap; check_problem only bounds them)
    bool levels_tile = true;
    for (int64_t l = 0, pre = 0; l < L; ++l) {
        levels_tile = levels_tile && pb.lsi[l] == pre;
        pre += pb.shapes[2 * l] * pb.shapes[2 * l + 1];
    }
    if (g_levelsum.load() && levels_tile) ls_levels = levelsum_levels<T>(pb);
    if (ls_levels != all_levels && (e = zero_grad_value()) != hipSuccess) return hip_fail(e, "zero-fill of grad_value");
    // Float atomics run at full rate only as >= 128-B row segments (one dword per lane): with 32 or more
    // channels put ONE channel on a lane, so that a wave-instruction adds two whole 128-B rows.
    int C = pick_channels_per_lane<T>(D, {value, grad_out});
    if (g_bwd_cpl.load() > 0) C = g_bwd_cpl.load() <= C ? g_bwd_cpl.load() : C;
    else if (D * (int)sizeof(T) >= 128 && ls_levels != all_levels) C = 1;
    msda::DirectGeom g = direct_geom(pb, C);
    const size_t lds = msda::direct_lds_bytes<T>(g);
    if (lds > 64 * 1024) return fail(MSDA_ERR_BAD_DIMS, "too many levels (L=%d) for the level table in LDS", L);
    const dim3 grid(direct_grid(g)), block(msda::kDirectThreads);
    ProfileScope prof(1, 1, (int)sizeof(T), N, S, M, D, L, Lq, P, stream);
    if (ls_levels) {
        // (forking this kernel onto a second stream beside the direct kernel was tried: the cross-stream fork / join costs
        // more than the overlap gains on a 90 us call -- 113 vs 96 us)
        e = launch_levelsum<T>(pb, loc, aw, grad_out, grad_value, stream, g.gv_skip);
        if (e != hipSuccess) return hip_fail(e, "launch of the level-sum backward kernel");
    }
    if constexpr (std::is_same<T, float>::value) {
        // grad_value is complete: the location / weight gradients of a small call come from the split kernel (bwd_split = 0 keeps the 8-lane one)
        if (g.gv_skip == all_levels && g_bwd_split.load() && split_fits(pb, C) && split_small(pb)) {
            e = launch_bwd_split<float>(pb, value, shapes, lsi, loc, aw, grad_out, grad_loc, grad_aw, g, stream);
            if (e != hipSuccess) return hip_fail(e, "launch of the split backward kernel");
            return MSDA_OK;
        }
    }
    switch (C) {
        case 4: hipLaunchKernelGGL((msda::bwd_direct_kernel<T, (sizeof(T) == 4 ? 4 : 2)>), grid, block, lds, stream, value, shapes, lsi, loc, aw, grad_out, grad_value, grad_loc, grad_aw, g); break;
        case 2: hipLaunchKernelGGL((msda::bwd_direct_kernel<T, 2>), grid, block, lds, stream, value, shapes, lsi, loc, aw, grad_out, grad_value, grad_loc, grad_aw, g); break;
        default: hipLaunchKernelGGL((msda::bwd_direct_kernel<T, 1>), grid, block, lds, stream, value, shapes, lsi, loc, aw, grad_out, grad_value, grad_loc, grad_aw, g); break;
    }
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the direct backward kernel");
    return MSDA_OK;
}

// ---- bf16 storage (value / out / grad_out / grad_value), fp32 compute ----------------------------------------------------------
// New capability: the reference dispatches float / double only (ms_deform_attn_cuda.cu:64,134).  Sampling locations,
// attention weights and their gradients stay fp32; every sum is formed in fp32 (or f64 LDS windows) and rounded to bf16 once.
bool bf16_scratch(hipStream_t stream, size_t n_floats, float **out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_ws_mu);
    Workspace &ws = g_ws[std::make_pair(dev, stream)];
    if (ws.gv32_cap < n_floats) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cap);
        if (cap != hipStreamCaptureStatusNone) return false;
        ws.retire(ws.gv32);      // (not freed: an earlier capture on this stream may hold the address)
        ws.gv32 = nullptr;
        ws.gv32_cap = 0;
        const size_t want = n_floats + n_floats / 2;
        if (hipMalloc(reinterpret_cast<void **>(&ws.gv32), want * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return false; }
        ws.gv32_cap = want;
    }
    *out = ws.gv32;
    return true;
}

// n4 vectors of four elements (dst 8-byte aligned), then the n - 4*n4 elements of the tail one by one
__global__ __launch_bounds__(256) void round_to_bf16_kernel(const float *__restrict__ src, msda::bf16_t *__restrict__ dst, size_t n4,
                                                             size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        msda::st4(dst + 4 * i, *reinterpret_cast<const float4 *>(src + 4 * i));
    for (size_t i = 4 * n4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = msda::to_storage<msda::bf16_t, float>(src[i]);
}

int pick_channels_bf16(int D, std::initializer_list<const void *> ptrs)
{
    int c = 4;
    for (; c > 1; c >>= 1) {
        bool ok = D % c == 0;
        for (const void *p : ptrs) ok = ok && is_aligned(p, 2 * (size_t)c);
        if (ok) break;
    }
    return c;
}

int forward_bf16_impl(const msda::bf16_t *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *aw,
                      int N, int S, int M, int D, int L, int Lq, int P, int im2col_step, msda::bf16_t *out,
                      const int64_t *shapes_host, const int64_t *lsi_host, msda_stream_t stream_)
{
    g_err[0] = 0;
    if (!value || !shapes || !lsi || !loc || !aw || !out) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Problem pb{N, S, M, D, L, Lq, P, {}, {}};
    if (int rc = check_problem(pb, shapes, lsi, shapes_host, lsi_host, im2col_step, stream)) return rc;
    if (!is_aligned(value, 2) || !is_aligned(out, 2) || !is_aligned(aw, 4) || !is_aligned(loc, 8) || !is_aligned(shapes, 8) ||
        !is_aligned(lsi, 8))
        return fail(MSDA_ERR_MISALIGNED, "misaligned pointer (sampling_loc needs 8 bytes)");

    int variant = g_tl_fwd_variant >= 0 ? g_tl_fwd_variant : g_fwd_variant.load();
    if (variant != 1 && is_aligned(value, 8) && is_aligned(out, 8) &&
        msda::plan_gather(N, S, M, D, L, Lq, P, pb.shapes.data(), pb.lsi.data()).ok) {
        unsigned *probe = nullptr;
        Monitor *mo = nullptr;
        if (variant == 0) {
            mo = monitor_for_current_device();
            variant = monitor_choose_fwd(mo, problem_key(N, S, M, L, P, pb.shapes.data(), loc), stream, &probe);
        }
        if (variant == 2) {
            hipError_t e;
            {
                ProfileScope prof(0, 2, 2, N, S, M, D, L, Lq, P, stream);
                e = msda::launch_fwd_tiled_tv<msda::bf16_t>(value, loc, aw, out, N, S, M, D, L, Lq, P, pb.shapes.data(),
                                                           pb.lsi.data(), probe, stream);
            }
            if (probe) monitor_finish_probe(mo, 2.0 * N * Lq * M * L * P, stream, e == hipSuccess);
            if (e != hipSuccess) return hip_fail(e, "launch of the tiled forward kernel (bf16)");
            return MSDA_OK;
        }
    }
    const int C = pick_channels_bf16(D, {value, out});
    const msda::DirectGeom g = direct_geom(pb, C);
    const size_t lds = msda::direct_lds_bytes<float>(g);
    if (lds > 64 * 1024) return fail(MSDA_ERR_BAD_DIMS, "too many levels (L=%d) for the level table in LDS", L);
    const dim3 grid(direct_grid(g)), block(msda::kDirectThreads);
    if (split_fits(pb, C) && (variant == 3 || (variant != 1 && split_small(pb)))) {
        ProfileScope prof(0, 3, 2, N, S, M, D, L, Lq, P, stream);
        const hipError_t e = launch_fwd_split<msda::bf16_t>(pb, value, shapes, lsi, loc, aw, out, g, stream);
        if (e != hipSuccess) return hip_fail(e, "launch of the split forward kernel (bf16)");
        return MSDA_OK;
    }
    ProfileScope prof(0, 1, 2, N, S, M, D, L, Lq, P, stream);
    const bool many = (int64_t)N * Lq * M >= 65536;
#define MSDA_LAUNCH_FWD(CC)                                                                                                          \
    if (many) hipLaunchKernelGGL((msda::fwd_direct_kernel<float, CC, 6, msda::bf16_t>), grid, block, lds, stream, value, shapes, lsi, loc, aw, out, g); \
    else hipLaunchKernelGGL((msda::fwd_direct_kernel<float, CC, 4, msda::bf16_t>), grid, block, lds, stream, value, shapes, lsi, loc, aw, out, g)
    switch (C) {
        case 4: MSDA_LAUNCH_FWD(4); break;
        case 2: MSDA_LAUNCH_FWD(2); break;
        default: MSDA_LAUNCH_FWD(1); break;
    }
#undef MSDA_LAUNCH_FWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the direct forward kernel (bf16)");
    return MSDA_OK;
}

int backward_bf16_impl(const msda::bf16_t *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *aw,
                       const msda::bf16_t *grad_out, int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                       msda::bf16_t *grad_value, float *grad_loc, float *grad_aw, const int64_t *shapes_host,
                       const int64_t *lsi_host, msda_stream_t stream_)
{
    g_err[0] = 0;
    if (!value || !shapes || !lsi || !loc || !aw || !grad_out || !grad_value || !grad_loc || !grad_aw)
        return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Problem pb{N, S, M, D, L, Lq, P, {}, {}};
    if (int rc = check_problem(pb, shapes, lsi, shapes_host, lsi_host, im2col_step, stream)) return rc;
    if (!is_aligned(value, 2) || !is_aligned(grad_out, 2) || !is_aligned(grad_value, 2) || !is_aligned(aw, 4) ||
        !is_aligned(grad_aw, 4) || !is_aligned(loc, 8) || !is_aligned(grad_loc, 8) || !is_aligned(shapes, 8) || !is_aligned(lsi, 8))
        return fail(MSDA_ERR_MISALIGNED, "misaligned pointer (sampling_loc / grad_sampling_loc need 8 bytes)");
    const size_t n_value = (size_t)N * S * M * D;
    hipError_t e = hipSuccess;
    auto finish_from_scratch = [&](float *gv32) {   // one rounding of the fp32 sums
        const size_t n4 = is_aligned(grad_value, 8) ? n_value / 4 : 0;
        hipLaunchKernelGGL(round_to_bf16_kernel, dim3(2048), dim3(256), 0, stream, gv32, grad_value, n4, n_value);
        return hipGetLastError();
    };

    int variant = g_bwd_variant.load();
    if (variant != 1 && is_aligned(value, 8) && is_aligned(grad_out, 8) && is_aligned(grad_value, 8)) {
        // automatic: encoder-shaped calls take the routed kernels (as in fp32; the rows they request per point are half as wide);
        // (5: the row-band kernel, a measured option -- see backward_impl)
        if (variant == 0) variant = Lq == S ? 4 : 1;
        float *gv32 = nullptr;
        if (variant == 5 && D == msda::kBandD) {
            const msda::BandPlan bp = msda::plan_band(N, S, M, D, L, Lq, P, pb.shapes.data(), pb.lsi.data());
            if (bp.ok && (!bp.atomic_levels || bf16_scratch(stream, n_value, &gv32))) {
                {
                    ProfileScope prof(1, 5, 2, N, S, M, D, L, Lq, P, stream);
                    e = launch_bwd_band<msda::bf16_t>(pb, value, loc, aw, grad_out, grad_value, gv32, grad_loc, grad_aw, stream);
                    if (e == hipErrorNotSupported) prof.cancel();
                }
                if (e == hipSuccess) return MSDA_OK;
                if (e != hipErrorNotSupported) return hip_fail(e, "launch of the row-band backward kernel (bf16)");
            }
            gv32 = nullptr;
        }
        if (variant == 4 && D == msda::kRpsD && bf16_scratch(stream, n_value, &gv32)) {
            // routed kernels: plain bf16 stores for the levels a workgroup owns alone; the levels shared by several workgroups are
            // accumulated in the fp32 scratch (zeroed by the route pass) and rounded once
            {
                ProfileScope prof(1, 4, 2, N, S, M, D, L, Lq, P, stream);
                e = launch_bwd_rps<msda::bf16_t>(pb, value, loc, aw, grad_out, grad_value, gv32, grad_loc, grad_aw, stream);
                if (e == hipErrorNotSupported) prof.cancel();
            }
            if (e == hipSuccess) return MSDA_OK;
            if (e != hipErrorNotSupported) return hip_fail(e, "launch of the routed backward kernels (bf16)");
        }
    }

    // direct path.  All levels summed in f64 LDS windows (msda_levelsum.h) when the plan allows: no atomics, no scratch,
    // the window is rounded to bf16 at its one store.  Otherwise every corner goes to an fp32 scratch buffer with row atomics.
    bool levels_tile = true;
    for (int64_t l = 0, pre = 0; l < L; ++l) {
        levels_tile = levels_tile && pb.lsi[l] == pre;
        pre += pb.shapes[2 * l] * pb.shapes[2 * l + 1];
    }
    const unsigned all_levels = L >= 32 ? ~0u : (1u << L) - 1;
    msda::LevelSumGeom lg;
    size_t ls_lds = 0;
    unsigned ls_levels = 0;
    if (g_levelsum.load() && levels_tile)
        ls_levels = msda::plan_levelsum(N, S, M, D, L, Lq, P, pb.shapes.data(), pb.lsi.data(), lg, ls_lds);
    const bool by_levelsum = ls_levels == all_levels;
    float *gv32 = nullptr;
    if (!by_levelsum) {
        if (!bf16_scratch(stream, n_value, &gv32))
            return fail(MSDA_ERR_BAD_DIMS, "bf16 backward of this shape needs an fp32 scratch buffer, which cannot be allocated "
                                           "while the stream is being captured: run the call once outside the capture");
        if ((e = hipMemsetAsync(gv32, 0, sizeof(float) * n_value, stream)) != hipSuccess) return hip_fail(e, "zero-fill of the fp32 scratch");
    }
    int C = pick_channels_bf16(D, {value, grad_out});
    if (!by_levelsum && D * 4 >= 128) C = 1;   // row atomics: one channel per lane (see backward_impl)
    msda::DirectGeom g = direct_geom(pb, C);
    const size_t lds = msda::direct_lds_bytes<float>(g);
    if (lds > 64 * 1024) return fail(MSDA_ERR_BAD_DIMS, "too many levels (L=%d) for the level table in LDS", L);
    const dim3 grid(direct_grid(g)), block(msda::kDirectThreads);
    ProfileScope prof(1, 1, 2, N, S, M, D, L, Lq, P, stream);
    if (by_levelsum) {
        const bool vec = P == 4 && is_aligned(loc, 16) && is_aligned(aw, 16);
        auto kern = vec ? &msda::bwd_levelsum_kernel<true, msda::bf16_t> : &msda::bwd_levelsum_kernel<false, msda::bf16_t>;
        if ((e = msda::set_lds_limit(reinterpret_cast<const void *>(kern), ls_lds)) != hipSuccess) return hip_fail(e, "LDS limit");
        hipLaunchKernelGGL(kern, dim3(msda::levelsum_grid(lg)), dim3(msda::kLsThreads), ls_lds, stream, loc, aw, grad_out, grad_value, lg);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "launch of the level-sum backward kernel (bf16)");
        g.gv_skip = all_levels;
        if (g_bwd_split.load() && split_fits(pb, C) && split_small(pb)) {
            e = launch_bwd_split<msda::bf16_t>(pb, value, shapes, lsi, loc, aw, grad_out, grad_loc, grad_aw, g, stream);
            if (e != hipSuccess) return hip_fail(e, "launch of the split backward kernel (bf16)");
            return MSDA_OK;
        }
    }
    switch (C) {
        case 4: hipLaunchKernelGGL((msda::bwd_direct_kernel<float, 4, msda::bf16_t>), grid, block, lds, stream, value, shapes, lsi, loc, aw, grad_out, gv32, grad_loc, grad_aw, g); break;
        case 2: hipLaunchKernelGGL((msda::bwd_direct_kernel<float, 2, msda::bf16_t>), grid, block, lds, stream, value, shapes, lsi, loc, aw, grad_out, gv32, grad_loc, grad_aw, g); break;
        default: hipLaunchKernelGGL((msda::bwd_direct_kernel<float, 1, msda::bf16_t>), grid, block, lds, stream, value, shapes, lsi, loc, aw, grad_out, gv32, grad_loc, grad_aw, g); break;
    }
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "launch of the direct backward kernel (bf16)");
    if (!by_levelsum) {
        e = finish_from_scratch(gv32);
        if (e != hipSuccess) return hip_fail(e, "rounding grad_value to bf16");
    }
    return MSDA_OK;
}

// ---- module-level element-wise kernels (msda_prep.h) -----------------------------------------------------------------------------
int prep_geom(msda::PrepGeom &g, int N, int Lq, int M, int L, int P, int ref_dim, const int64_t *shapes_host)
{
    if (N <= 0 || Lq <= 0 || M <= 0 || L <= 0 || P <= 0) return fail(MSDA_ERR_BAD_DIMS, "non-positive dimension");
    if (L > msda::kPrepMaxL || L * P > 64) return fail(MSDA_ERR_BAD_DIMS, "L (%d) > %d or L*P (%d) > 64", L, msda::kPrepMaxL, L * P);
    if (ref_dim != 2 && ref_dim != 4) return fail(MSDA_ERR_BAD_DIMS, "Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
    if (!shapes_host) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    g = msda::PrepGeom{};
    g.N = N; g.Lq = Lq; g.M = M; g.L = L; g.P = P; g.ref_dim = ref_dim;
    int G = 1;
    while (G < L * P) G <<= 1;
    g.G = G;
    for (int l = 0; l < L; ++l) {
        g.H[l] = (float)shapes_host[2 * l];
        g.W[l] = (float)shapes_host[2 * l + 1];
    }
    return MSDA_OK;
}

template <typename T>
int roi_align_impl(const T *input, const T *rois, int K, int N, int C, int H, int W, int PH, int PW, double spatial_scale,
                   int sampling_ratio, int aligned, T *output, msda_stream_t stream)
{
    g_err[0] = 0;
    if (!input || !rois || !output) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    if (K < 0 || N < 1 || C < 1 || H < 1 || W < 1 || PH < 1 || PW < 1 || sampling_ratio < 0 || !(spatial_scale > 0))
        return fail(MSDA_ERR_BAD_DIMS, "bad ROIAlign dimensions");
    const int64_t n_out = (int64_t)K * C * PH * PW;
    if (n_out == 0) return MSDA_OK;
    if ((int64_t)N * C * H * W >= ((int64_t)1 << 40)) return fail(MSDA_ERR_TOO_LARGE, "input too large");
    const int grid = (int)std::min<int64_t>((n_out + 255) / 256, 65536);
    hipLaunchKernelGGL(msda::roi_align_fwd_kernel<T>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), input, rois, n_out, N, C,
                       H, W, PH, PW, (T)spatial_scale, sampling_ratio, aligned, output);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the ROIAlign kernel");
    return MSDA_OK;
}

template <typename T, typename TP = T>
int prep_forward_impl(const TP *offsets, int64_t off_stride, const TP *logits, int64_t log_stride, const T *ref, int ref_dim,
                      const int64_t *shapes_host, int N, int Lq, int M, int L, int P, T *loc, T *aw, msda_stream_t stream_)
{
    g_err[0] = 0;
    if (!offsets || !logits || !ref || !loc || !aw) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    msda::PrepGeom g;
    if (int rc = prep_geom(g, N, Lq, M, L, P, ref_dim, shapes_host)) return rc;
    if (off_stride < (int64_t)M * L * P * 2 || log_stride < (int64_t)M * L * P)
        return fail(MSDA_ERR_BAD_DIMS, "row stride smaller than a row (offsets %lld, logits %lld)", (long long)off_stride, (long long)log_stride);
    g.off_stride = off_stride;
    g.log_stride = log_stride;
    const int64_t items = (int64_t)N * Lq * M, per_block = 256 / g.G;
    const int grid = (int)std::min<int64_t>((items + per_block - 1) / per_block, 8192);
    hipLaunchKernelGGL((msda::prep_forward_kernel<T, TP>), dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream_), offsets, logits, ref, loc, aw, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the location / softmax kernel");
    return MSDA_OK;
}

template <typename T, typename TP = T>
int prep_backward_impl(const T *grad_loc, const T *grad_aw, const T *aw, const TP *offsets, int64_t off_stride, const T *ref,
                       int ref_dim, const int64_t *shapes_host, int N, int Lq, int M, int L, int P, TP *grad_offsets,
                       int64_t goff_stride, TP *grad_logits, int64_t glog_stride, T *grad_ref, msda_stream_t stream_)
{
    g_err[0] = 0;
    if (!grad_loc || !grad_aw || !aw || !ref || !grad_offsets || !grad_logits || (ref_dim == 4 && grad_ref && !offsets))
        return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    msda::PrepGeom g;
    if (int rc = prep_geom(g, N, Lq, M, L, P, ref_dim, shapes_host)) return rc;
    if (goff_stride < (int64_t)M * L * P * 2 || glog_stride < (int64_t)M * L * P)
        return fail(MSDA_ERR_BAD_DIMS, "row stride smaller than a row");
    g.off_stride = off_stride;
    g.goff_stride = goff_stride;
    g.glog_stride = glog_stride;
    const int64_t items = (int64_t)N * Lq, per_block = 256 / g.G;
    const int grid = (int)std::min<int64_t>((items + per_block - 1) / per_block, 8192);
    hipLaunchKernelGGL((msda::prep_backward_kernel<T, TP>), dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream_), grad_loc, grad_aw,
                       aw, offsets, ref, grad_offsets, grad_logits, grad_ref, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the location / softmax backward kernel");
    return MSDA_OK;
}

template <typename T>
int mask_rows_impl(T *x, const uint8_t *mask, int64_t rows, int row_elems, msda_stream_t stream_)
{
    g_err[0] = 0;
    if (!x || !mask) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    if (rows <= 0 || row_elems <= 0) return fail(MSDA_ERR_BAD_DIMS, "non-positive dimension");
    const int grid = (int)std::min<int64_t>((rows + 255) / 256, 4096);
    hipLaunchKernelGGL(msda::mask_rows_kernel<T>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream_), x, mask, (long long)rows, row_elems);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the padding-mask kernel");
    return MSDA_OK;
}

}  // namespace

// ---- msda_forward_prep_*: the module's softmax + location arithmetic and the operator's forward behind ONE entry point ----------------
// Decoder-shaped calls (Lq != S) with L * P <= 32 run fwd_direct_prep_kernel -- one launch, sampling_loc / attn_weight written as a
// by-product --; everything else (the encoder-shaped calls keep their LDS-window kernel and its locality monitor, which read
// sampling_loc) runs the location / softmax kernel and then the forward of the same library, as two launches.  Results are those of
// msda_prep_forward_* followed by msda_forward_* (the softmax sums in a different order: last-ulp differences).
template <typename T, typename TV, typename TP>
int forward_prep_impl(const TV *value, const int64_t *shapes, const int64_t *lsi, const TP *offsets, int64_t off_stride, const TP *logits,
                      int64_t log_stride, const T *ref, int ref_dim, int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                      TV *out, T *loc, T *aw, const int64_t *shapes_host, const int64_t *lsi_host, msda_stream_t stream_)
{
    g_err[0] = 0;
    if (!value || !shapes || !lsi || !offsets || !logits || !ref || !out || !loc || !aw) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    if (ref_dim != 2 && ref_dim != 4) return fail(MSDA_ERR_BAD_DIMS, "reference points must have 2 or 4 components, got %d", ref_dim);
    if (off_stride < (int64_t)M * L * P * 2 || log_stride < (int64_t)M * L * P)
        return fail(MSDA_ERR_BAD_DIMS, "row stride smaller than a row (offsets %lld, logits %lld)", (long long)off_stride, (long long)log_stride);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool fused = g_fwd_prep_fused.load() && Lq != S && L * P <= msda::kPointBatch;
    // Encoder-shaped calls (round 5): the LDS-window kernel reads the raw projection itself -- softmax by the
    // quad that owns the query, location arithmetic by the lane that resolves the point -- and writes sampling_loc / attn_weight as
    // by-products: no prep_forward_kernel launch, no re-read of the two tensors.  While the locality monitor sends the call to the direct
    // kernel (spread sampling points) the two-kernel form below runs, with that choice held.
    if constexpr (std::is_same<T, float>::value) {
        if (g_fwd_prep_fused.load() >= 2 && !fused && P == 4 && g_fwd_variant.load() != 1) {
            Problem pb{N, S, M, D, L, Lq, P, {}, {}};
            if (int rc = check_problem(pb, shapes, lsi, shapes_host, lsi_host, im2col_step, stream)) return rc;
            constexpr size_t row_align = sizeof(TV) * 4;
            const bool aligned = is_aligned(value, row_align) && is_aligned(out, row_align) && is_aligned(offsets, 2 * sizeof(TP)) && off_stride % 2 == 0 &&
                                 is_aligned(logits, sizeof(TP)) && is_aligned(ref, 8) && is_aligned(loc, 8) && is_aligned(aw, 4);
            if (aligned && msda::plan_gather(N, S, M, D, L, Lq, P, pb.shapes.data(), pb.lsi.data()).ok) {
                int variant = g_fwd_variant.load();
                unsigned *probe = nullptr;
                Monitor *mo = nullptr;
                if (variant == 0) {
                    mo = monitor_for_current_device();
                    variant = monitor_choose_fwd(mo, problem_key(N, S, M, L, P, pb.shapes.data(), loc), stream, &probe);
                }
                if (variant == 2) {
                    const msda::TiledPrepSrc src{offsets, logits, (long long)off_stride, (long long)log_stride, ref, ref_dim, loc, aw};
                    hipError_t e;
                    {
                        ProfileScope prof(0, 6, (int)sizeof(TV), N, S, M, D, L, Lq, P, stream);
                        e = msda::launch_fwd_tiled_prep<TV, TP>(value, src, out, N, S, M, D, L, Lq, P, pb.shapes.data(), pb.lsi.data(), probe, stream);
                    }
                    if (probe) monitor_finish_probe(mo, 2.0 * N * Lq * M * L * P, stream, e == hipSuccess);
                    if (e != hipSuccess) return hip_fail(e, "launch of the fused location / softmax / window-gather kernel");
                    return MSDA_OK;
                }
                // the monitor chose the direct kernel: location / softmax kernel, then the direct forward (the choice is not asked for again)
                if (int rc = prep_forward_impl<T, TP>(offsets, off_stride, logits, log_stride, ref, ref_dim, shapes_host, N, Lq, M, L, P, loc, aw, stream_))
                    return rc;
                g_tl_fwd_variant = 1;
                int rc;
                if constexpr (std::is_same<TV, msda::bf16_t>::value)
                    rc = forward_bf16_impl(value, shapes, lsi, loc, aw, N, S, M, D, L, Lq, P, im2col_step, out, shapes_host, lsi_host, stream_);
                else
                    rc = forward_impl<T>(value, shapes, lsi, loc, aw, N, S, M, D, L, Lq, P, im2col_step, out, shapes_host, lsi_host, stream_);
                g_tl_fwd_variant = -1;
                return rc;
            }
        }
    }
    if (!fused) {
        if (int rc = prep_forward_impl<T, TP>(offsets, off_stride, logits, log_stride, ref, ref_dim, shapes_host, N, Lq, M, L, P, loc, aw, stream_))
            return rc;
        if constexpr (std::is_same<TV, msda::bf16_t>::value)
            return forward_bf16_impl(value, shapes, lsi, loc, aw, N, S, M, D, L, Lq, P, im2col_step, out, shapes_host, lsi_host, stream_);
        else
            return forward_impl<T>(value, shapes, lsi, loc, aw, N, S, M, D, L, Lq, P, im2col_step, out, shapes_host, lsi_host, stream_);
    }
    Problem pb{N, S, M, D, L, Lq, P, {}, {}};
    if (int rc = check_problem(pb, shapes, lsi, shapes_host, lsi_host, im2col_step, stream)) return rc;
    if (!is_aligned(value, sizeof(TV)) || !is_aligned(out, sizeof(TV)) || !is_aligned(aw, sizeof(T)) || !is_aligned(loc, 2 * sizeof(T)) ||
        !is_aligned(ref, sizeof(T)) || !is_aligned(offsets, sizeof(TP)) || !is_aligned(logits, sizeof(TP)) || !is_aligned(shapes, 8) || !is_aligned(lsi, 8))
        return fail(MSDA_ERR_MISALIGNED, "misaligned pointer (sampling_loc needs 2*sizeof(T))");
    int C;
    if constexpr (std::is_same<TV, msda::bf16_t>::value) C = pick_channels_bf16(D, {value, out});
    else C = pick_channels_per_lane<T>(D, {value, out});
    const msda::DirectGeom g = direct_geom(pb, C);
    const size_t lds = msda::direct_lds_bytes<T>(g);
    if (lds > 64 * 1024) return fail(MSDA_ERR_BAD_DIMS, "too many levels (L=%d) for the level table in LDS", L);
    const dim3 grid(direct_grid(g)), block(msda::kDirectThreads);
    const msda::PrepSrc<TP> src{offsets, logits, off_stride, log_stride, ref_dim};
    ProfileScope prof(0, 5, (int)sizeof(TV), N, S, M, D, L, Lq, P, stream);
    const bool many = (int64_t)N * Lq * M >= 65536;
#define MSDA_LAUNCH_FWDP(CC)                                                                                                                    \
    if (many) hipLaunchKernelGGL((msda::fwd_direct_prep_kernel<T, CC, 6, TV, TP>), grid, block, lds, stream, value, shapes, lsi, src, ref, loc, aw, out, g); \
    else hipLaunchKernelGGL((msda::fwd_direct_prep_kernel<T, CC, 4, TV, TP>), grid, block, lds, stream, value, shapes, lsi, src, ref, loc, aw, out, g)
    switch (C) {
        case 4: MSDA_LAUNCH_FWDP((sizeof(T) == 4 ? 4 : 2)); break;
        case 2: MSDA_LAUNCH_FWDP(2); break;
        default: MSDA_LAUNCH_FWDP(1); break;
    }
#undef MSDA_LAUNCH_FWDP
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the fused location / softmax / gather kernel");
    return MSDA_OK;
}

extern "C" {

int msda_abi_version(void) { return RICHSEM_MSDA_ABI_VERSION; }

// the host variants the reference declares and does not implement (src/cpu/ms_deform_attn_cpu.cpp:17-41)
int msda_forward_cpu(const void *, const int64_t *, const int64_t *, const void *, const void *, int, int, int, int, int, int, int, int, void *)
{
    return fail(MSDA_ERR_NOT_ON_CPU, "Not implement on cpu");
}
int msda_backward_cpu(const void *, const int64_t *, const int64_t *, const void *, const void *, const void *, int, int, int, int, int, int, int,
                      int, void *, void *, void *)
{
    return fail(MSDA_ERR_NOT_ON_CPU, "Not implement on cpu");
}

const char *msda_last_error(void) { return g_err; }

int msda_set_option(const char *key, int value)
{
    if (key && !strcmp(key, "fwd_variant") && value >= 0 && value <= 3) { g_fwd_variant = value; return MSDA_OK; }
    if (key && !strcmp(key, "fwd_prep_fused") && value >= 0 && value <= 2) { g_fwd_prep_fused = value; return MSDA_OK; }      // 1: decoder-shaped calls; 2: + encoder-shaped
    if (key && !strcmp(key, "bwd_variant") && (value == 0 || value == 1 || value == 4 || value == 5)) { g_bwd_variant = value; return MSDA_OK; }
    if (key && !strcmp(key, "band_lds_kb") && value >= 16 && value <= 150) { msda::band_options().lds_kb = value; return MSDA_OK; }
    if (key && !strcmp(key, "band_hits") && value >= 32 && value <= 65536) { msda::band_options().hits = value; return MSDA_OK; }
    if (key && !strcmp(key, "rps_tile") && value >= 4 && value <= 16) { msda::rps_options().tile = value; return MSDA_OK; }
    if (key && !strcmp(key, "rps_max_chunks") && value >= 1 && value <= 4096) { msda::rps_options().max_chunks = value; return MSDA_OK; }
    if (key && !strcmp(key, "rps_route_wgs") && value >= 1 && value <= 64) { msda::rps_options().route_wgs = value; return MSDA_OK; }
    if (key && !strcmp(key, "rps_seg_shift") && value >= 3 && value <= 11) { msda::rps_options().seg_shift = value; return MSDA_OK; }
    if (key && !strcmp(key, "rps_order") && value >= 0 && value <= 1) { msda::rps_options().order = value; return MSDA_OK; }
    if (key && !strcmp(key, "bwd_direct_cpl") && (value == 0 || value == 1 || value == 2 || value == 4)) { g_bwd_cpl = value; return MSDA_OK; }
    if (key && !strcmp(key, "tile_region") && value >= 4 && value <= 64) { msda::tiled_options().region_px = value; return MSDA_OK; }
    if (key && !strcmp(key, "tile_margin") && value >= 0 && value <= 32) { msda::tiled_options().margin = value; return MSDA_OK; }
    if (key && !strcmp(key, "tile_debug") && value >= 0 && value <= 65535) { msda::tiled_options().dbg = value; return MSDA_OK; }
    if (key && !strcmp(key, "tile_persist") && value >= 0 && value <= 65536) { msda::tiled_options().persist = value; return MSDA_OK; }
    if (key && !strcmp(key, "bwd_levelsum") && (value == 0 || value == 1)) { g_levelsum = value; return MSDA_OK; }
    if (key && !strcmp(key, "bwd_split") && (value == 0 || value == 1)) { g_bwd_split = value; return MSDA_OK; }
    if (key && !strcmp(key, "profile_filter") && value >= 0 && value <= 47) { g_prof_filter = value; return MSDA_OK; }
    if (key && !strcmp(key, "levelsum_lds_kb") && value >= 8 && value <= 150) { msda::levelsum_lds_kb() = value; return MSDA_OK; }
    if (key && !strcmp(key, "tile_grow") && (value == 0 || value == 1)) { msda::tiled_options().grow = value; return MSDA_OK; }
    if (key && !strcmp(key, "locality_monitor") && (value == 0 || value == 1)) {
        g_monitor_on = value;
        for (Monitor &mo : g_monitors) {   // switching it (either way) forgets what was learnt
            std::lock_guard<std::mutex> lock(mo.mu);
            mo.table.clear();
            for (ProbeSlot &ps : mo.slot) ps.pending = false;   // (probes still in flight belong to what is being forgotten)
        }
        g_last_share_ppm = -1;
        return MSDA_OK;
    }
    return fail(MSDA_ERR_BAD_OPTION, "unknown option or value: %s=%d", key ? key : "(null)", value);
}

int msda_get_option(const char *key, int *value)
{
    if (!value) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    if (key && !strcmp(key, "fwd_variant")) { *value = g_fwd_variant; return MSDA_OK; }
    if (key && !strcmp(key, "fwd_prep_fused")) { *value = g_fwd_prep_fused; return MSDA_OK; }
    if (key && !strcmp(key, "bwd_variant")) { *value = g_bwd_variant; return MSDA_OK; }
    if (key && !strcmp(key, "band_lds_kb")) { *value = msda::band_options().lds_kb; return MSDA_OK; }
    if (key && !strcmp(key, "band_hits")) { *value = msda::band_options().hits; return MSDA_OK; }
    if (key && !strcmp(key, "bwd_direct_cpl")) { *value = g_bwd_cpl; return MSDA_OK; }
    if (key && !strcmp(key, "rps_tile")) { *value = msda::rps_options().tile; return MSDA_OK; }
    if (key && !strcmp(key, "rps_max_chunks")) { *value = msda::rps_options().max_chunks; return MSDA_OK; }
    if (key && !strcmp(key, "rps_route_wgs")) { *value = msda::rps_options().route_wgs; return MSDA_OK; }
    if (key && !strcmp(key, "rps_seg_shift")) { *value = msda::rps_options().seg_shift; return MSDA_OK; }
    if (key && !strcmp(key, "rps_order")) { *value = msda::rps_options().order; return MSDA_OK; }
    if (key && !strcmp(key, "tile_region")) { *value = msda::tiled_options().region_px; return MSDA_OK; }
    if (key && !strcmp(key, "tile_margin")) { *value = msda::tiled_options().margin; return MSDA_OK; }
    if (key && !strcmp(key, "tile_persist")) { *value = msda::tiled_options().persist; return MSDA_OK; }
    if (key && !strcmp(key, "bwd_levelsum")) { *value = g_levelsum; return MSDA_OK; }
    if (key && !strcmp(key, "bwd_split")) { *value = g_bwd_split; return MSDA_OK; }
    if (key && !strcmp(key, "profile_filter")) { *value = g_prof_filter; return MSDA_OK; }
    if (key && !strcmp(key, "levelsum_lds_kb")) { *value = msda::levelsum_lds_kb(); return MSDA_OK; }
    if (key && !strcmp(key, "tile_grow")) { *value = msda::tiled_options().grow; return MSDA_OK; }
    if (key && !strcmp(key, "locality_monitor")) { *value = g_monitor_on; return MSDA_OK; }
    if (key && !strcmp(key, "locality_share_ppm")) { *value = g_last_share_ppm; return MSDA_OK; }   // read-only
    return fail(MSDA_ERR_BAD_OPTION, "unknown option: %s", key ? key : "(null)");
}

int msda_tiled_plan(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes_host,
                    const int64_t *level_start_host, int *info)
{
    if (!shapes_host || !level_start_host || !info) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    if (L < 1 || L > 64) return fail(MSDA_ERR_BAD_DIMS, "bad L=%d", L);
    const msda::TiledPlan pl = msda::plan_gather(N, S, M, D, L, Lq, P, shapes_host, level_start_host);
    info[0] = pl.ok ? 1 : 0;
    info[1] = pl.g.GY;
    info[2] = pl.g.GX;
    info[3] = pl.g.nphases;
    info[4] = (int)pl.lds_bytes;
    info[5] = pl.grid * (msda::kTD / msda::kFwdGC);
    info[6] = pl.g.margin;
    info[7] = 0;
    if (pl.ok) {   // largest number of queries any region holds
        for (int gy = 0; gy < pl.g.GY; ++gy)
            for (int gx = 0; gx < pl.g.GX; ++gx) {
                int nq = 0;
                for (int l = 0; l < L; ++l) {
                    const msda::LevelRect r = msda::level_rect(pl.g.H[l], pl.g.W[l], gy, gx, pl.g.GY, pl.g.GX, pl.g.margin_l[l]);
                    nq += r.qnr * r.qnc;
                }
                info[7] = nq > info[7] ? nq : info[7];
            }
    }
    return MSDA_OK;
}

int msda_levelsum_plan(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes_host,
                       const int64_t *level_start_host, int *info)
{
    if (!shapes_host || !level_start_host || !info) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    if (L < 1 || L > 64) return fail(MSDA_ERR_BAD_DIMS, "bad L=%d", L);
    msda::LevelSumGeom lg;
    size_t lds = 0;
    const unsigned mask = msda::plan_levelsum(N, S, M, D, L, Lq, P, shapes_host, level_start_host, lg, lds);
    info[0] = (int)mask;
    info[1] = lg.nlev;
    info[2] = lg.nslices;
    info[3] = (int)lds;
    info[4] = mask ? msda::levelsum_grid(lg) : 0;
    int rows = 0;
    for (int e = 0; e < lg.nlev; ++e) rows = lg.nr[e] > rows ? lg.nr[e] : rows;
    info[5] = rows;
    info[6] = info[7] = 0;
    return MSDA_OK;
}

int msda_debug_stats(void *device_counter)
{
    msda::tiled_options().stats = static_cast<unsigned *>(device_counter);
    return MSDA_OK;
}

int msda_debug_stamps(void *device_buffer)
{
    msda::tiled_options().stamps = static_cast<unsigned long long *>(device_buffer);
    return MSDA_OK;
}

int msda_profile_enable(int capacity)
{
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_on = false;
    for (auto &s : g_prof_slots) {
        (void)hipEventDestroy(s.start);
        (void)hipEventDestroy(s.stop);
    }
    g_prof_slots.clear();
    g_prof_used = 0;
    if (capacity <= 0) return MSDA_OK;
    g_prof_slots.resize((size_t)capacity);
    for (auto &s : g_prof_slots) {
        hipError_t e = hipEventCreate(&s.start);
        if (e == hipSuccess) e = hipEventCreate(&s.stop);
        if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
    }
    g_prof_on = true;
    return MSDA_OK;
}

int msda_profile_collect(msda_profile_record *records, int max_records, int *n_records)
{
    if (!records || !n_records) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    int n = 0;
    for (int i = 0; i < g_prof_used && n < max_records; ++i) {
        ProfileSlot &s = g_prof_slots[i];
        hipError_t e = hipEventSynchronize(s.stop);
        if (e == hipSuccess) e = hipEventElapsedTime(&s.rec.kernel_ms, s.start, s.stop);
        if (e != hipSuccess) return hip_fail(e, "reading profile events");
        records[n++] = s.rec;
    }
    *n_records = n;
    g_prof_used = 0;
    return MSDA_OK;
}

int msda_forward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const float *sampling_loc, const float *attn_weight, int N, int S, int M, int D, int L, int Lq,
                     int P, int im2col_step, float *out, const int64_t *shapes_host, const int64_t *level_start_host,
                     msda_stream_t stream)
{
    return forward_impl<float>(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L, Lq, P,
                               im2col_step, out, shapes_host, level_start_host, stream);
}

int msda_forward_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const double *sampling_loc, const double *attn_weight, int N, int S, int M, int D, int L, int Lq,
                     int P, int im2col_step, double *out, const int64_t *shapes_host, const int64_t *level_start_host,
                     msda_stream_t stream)
{
    return forward_impl<double>(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L, Lq, P,
                                im2col_step, out, shapes_host, level_start_host, stream);
}

int msda_backward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                      const float *sampling_loc, const float *attn_weight, const float *grad_out, int N, int S, int M,
                      int D, int L, int Lq, int P, int im2col_step, float *grad_value, float *grad_sampling_loc,
                      float *grad_attn_weight, const int64_t *shapes_host, const int64_t *level_start_host,
                      msda_stream_t stream)
{
    return backward_impl<float>(value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_out, N, S, M, D, L,
                                Lq, P, im2col_step, grad_value, grad_sampling_loc, grad_attn_weight, shapes_host,
                                level_start_host, stream);
}

int msda_backward_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                      const double *sampling_loc, const double *attn_weight, const double *grad_out, int N, int S,
                      int M, int D, int L, int Lq, int P, int im2col_step, double *grad_value,
                      double *grad_sampling_loc, double *grad_attn_weight, const int64_t *shapes_host,
                      const int64_t *level_start_host, msda_stream_t stream)
{
    return backward_impl<double>(value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_out, N, S, M, D, L,
                                 Lq, P, im2col_step, grad_value, grad_sampling_loc, grad_attn_weight, shapes_host,
                                 level_start_host, stream);
}

int msda_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                      const float *sampling_loc, const float *attn_weight, int N, int S, int M, int D, int L, int Lq, int P,
                      int im2col_step, uint16_t *out, const int64_t *shapes_host, const int64_t *level_start_host,
                      msda_stream_t stream)
{
    return forward_bf16_impl(reinterpret_cast<const msda::bf16_t *>(value), spatial_shapes, level_start, sampling_loc,
                             attn_weight, N, S, M, D, L, Lq, P, im2col_step, reinterpret_cast<msda::bf16_t *>(out), shapes_host,
                             level_start_host, stream);
}

int msda_backward_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                       const float *sampling_loc, const float *attn_weight, const uint16_t *grad_out, int N, int S, int M,
                       int D, int L, int Lq, int P, int im2col_step, uint16_t *grad_value, float *grad_sampling_loc,
                       float *grad_attn_weight, const int64_t *shapes_host, const int64_t *level_start_host,
                       msda_stream_t stream)
{
    return backward_bf16_impl(reinterpret_cast<const msda::bf16_t *>(value), spatial_shapes, level_start, sampling_loc,
                              attn_weight, reinterpret_cast<const msda::bf16_t *>(grad_out), N, S, M, D, L, Lq, P, im2col_step,
                              reinterpret_cast<msda::bf16_t *>(grad_value), grad_sampling_loc, grad_attn_weight, shapes_host,
                              level_start_host, stream);
}

int msda_forward_prep_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start, const float *offsets,
                          int64_t off_stride, const float *logits, int64_t log_stride, const float *ref, int ref_dim, int N, int S, int M, int D,
                          int L, int Lq, int P, int im2col_step, float *out, float *sampling_loc, float *attn_weight,
                          const int64_t *shapes_host, const int64_t *level_start_host, msda_stream_t stream)
{
    return forward_prep_impl<float, float, float>(value, spatial_shapes, level_start, offsets, off_stride, logits, log_stride, ref, ref_dim, N, S, M,
                                                  D, L, Lq, P, im2col_step, out, sampling_loc, attn_weight, shapes_host, level_start_host, stream);
}
int msda_forward_prep_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start, const double *offsets,
                          int64_t off_stride, const double *logits, int64_t log_stride, const double *ref, int ref_dim, int N, int S, int M, int D,
                          int L, int Lq, int P, int im2col_step, double *out, double *sampling_loc, double *attn_weight,
                          const int64_t *shapes_host, const int64_t *level_start_host, msda_stream_t stream)
{
    return forward_prep_impl<double, double, double>(value, spatial_shapes, level_start, offsets, off_stride, logits, log_stride, ref, ref_dim, N, S,
                                                     M, D, L, Lq, P, im2col_step, out, sampling_loc, attn_weight, shapes_host, level_start_host,
                                                     stream);
}
int msda_forward_prep_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start, const uint16_t *offsets,
                           int64_t off_stride, const uint16_t *logits, int64_t log_stride, const float *ref, int ref_dim, int N, int S, int M,
                           int D, int L, int Lq, int P, int im2col_step, uint16_t *out, float *sampling_loc, float *attn_weight,
                           const int64_t *shapes_host, const int64_t *level_start_host, msda_stream_t stream)
{
    return forward_prep_impl<float, msda::bf16_t, msda::bf16_t>(
        reinterpret_cast<const msda::bf16_t *>(value), spatial_shapes, level_start, reinterpret_cast<const msda::bf16_t *>(offsets), off_stride,
        reinterpret_cast<const msda::bf16_t *>(logits), log_stride, ref, ref_dim, N, S, M, D, L, Lq, P, im2col_step,
        reinterpret_cast<msda::bf16_t *>(out), sampling_loc, attn_weight, shapes_host, level_start_host, stream);
}

#define MSDA_PREP_EXPORTS(SFX, T)                                                                                                  \
    int msda_prep_forward_##SFX(const T *offsets, int64_t off_stride, const T *logits, int64_t log_stride, const T *ref, int ref_dim, \
                                const int64_t *shapes_host, int N, int Lq, int M, int L, int P, T *loc, T *aw, msda_stream_t stream)  \
    {                                                                                                                             \
        return prep_forward_impl<T>(offsets, off_stride, logits, log_stride, ref, ref_dim, shapes_host, N, Lq, M, L, P, loc, aw,    \
                                    stream);                                                                                      \
    }                                                                                                                             \
    int msda_prep_backward_##SFX(const T *grad_loc, const T *grad_aw, const T *aw, const T *offsets, int64_t off_stride,           \
                                 const T *ref, int ref_dim, const int64_t *shapes_host, int N, int Lq, int M, int L, int P,         \
                                 T *grad_offsets, int64_t goff_stride, T *grad_logits, int64_t glog_stride, T *grad_ref,            \
                                 msda_stream_t stream)                                                                            \
    {                                                                                                                             \
        return prep_backward_impl<T>(grad_loc, grad_aw, aw, offsets, off_stride, ref, ref_dim, shapes_host, N, Lq, M, L, P,         \
                                     grad_offsets, goff_stride, grad_logits, glog_stride, grad_ref, stream);                       \
    }
MSDA_PREP_EXPORTS(f32, float)
MSDA_PREP_EXPORTS(f64, double)
#undef MSDA_PREP_EXPORTS

int msda_prep_forward_bf16(const uint16_t *offsets, int64_t off_stride, const uint16_t *logits, int64_t log_stride, const float *ref,
                           int ref_dim, const int64_t *shapes_host, int N, int Lq, int M, int L, int P, float *loc, float *aw,
                           msda_stream_t stream)
{
    return prep_forward_impl<float, msda::bf16_t>(reinterpret_cast<const msda::bf16_t *>(offsets), off_stride,
                                                  reinterpret_cast<const msda::bf16_t *>(logits), log_stride, ref, ref_dim, shapes_host, N,
                                                  Lq, M, L, P, loc, aw, stream);
}
int msda_prep_backward_bf16(const float *grad_loc, const float *grad_aw, const float *aw, const uint16_t *offsets, int64_t off_stride,
                            const float *ref, int ref_dim, const int64_t *shapes_host, int N, int Lq, int M, int L, int P,
                            uint16_t *grad_offsets, int64_t goff_stride, uint16_t *grad_logits, int64_t glog_stride, float *grad_ref,
                            msda_stream_t stream)
{
    return prep_backward_impl<float, msda::bf16_t>(grad_loc, grad_aw, aw, reinterpret_cast<const msda::bf16_t *>(offsets), off_stride, ref,
                                                   ref_dim, shapes_host, N, Lq, M, L, P, reinterpret_cast<msda::bf16_t *>(grad_offsets),
                                                   goff_stride, reinterpret_cast<msda::bf16_t *>(grad_logits), glog_stride, grad_ref, stream);
}

int msda_dn_indices_i64(const int64_t *cum, int batch, int64_t total, int groups2, int64_t single_pad, int64_t *known_bid,
                        int64_t *map_known_indice, msda_stream_t stream)
{
    g_err[0] = 0;
    if (!cum || !known_bid || !map_known_indice) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    if (batch < 1 || total < 0 || groups2 < 0 || single_pad < 0) return fail(MSDA_ERR_BAD_DIMS, "bad denoising dimensions");
    const int64_t n = total * groups2;
    if (n == 0) return MSDA_OK;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(msda::dn_indices_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), cum, batch, total, n, single_pad,
                       known_bid, map_known_indice);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the denoising index kernel");
    return MSDA_OK;
}

int msda_dn_attn_mask_u8(uint8_t *mask, int64_t tgt_size, int64_t pad_size, int64_t group_pad, msda_stream_t stream)
{
    g_err[0] = 0;
    if (!mask) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    if (tgt_size < 0 || pad_size < 0 || pad_size > tgt_size || group_pad < 0) return fail(MSDA_ERR_BAD_DIMS, "bad mask dimensions");
    if (tgt_size == 0) return MSDA_OK;
    const int64_t n = tgt_size * tgt_size;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(msda::dn_attn_mask_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), mask, tgt_size, pad_size,
                       group_pad);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the denoising mask kernel");
    return MSDA_OK;
}

int msda_roi_align_forward_f32(const float *input, const float *rois, int K, int N, int C, int H, int W, int pooled_h, int pooled_w,
                               double spatial_scale, int sampling_ratio, int aligned, float *output, msda_stream_t stream)
{
    return roi_align_impl<float>(input, rois, K, N, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, output, stream);
}
int msda_roi_align_forward_f64(const double *input, const double *rois, int K, int N, int C, int H, int W, int pooled_h, int pooled_w,
                               double spatial_scale, int sampling_ratio, int aligned, double *output, msda_stream_t stream)
{
    return roi_align_impl<double>(input, rois, K, N, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, output, stream);
}

int msda_topk_f32(const float *scores, int rows, int n, int k, int64_t *indices, float *values, msda_stream_t stream)
{
    g_err[0] = 0;
    if (!scores || !indices) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
    if (rows < 0 || n < 1 || k < 1 || k > n || k > msda::kTopkMaxK || n > msda::kTopkMaxN)
        return fail(MSDA_ERR_BAD_DIMS, "top-k: 1 <= k <= min(n, %d), n <= %d (got n=%d, k=%d)", msda::kTopkMaxK, msda::kTopkMaxN, n, k);
    if (rows == 0) return MSDA_OK;
    const size_t lds = msda::topk_lds_bytes(n);
    hipError_t e = msda::set_lds_limit(reinterpret_cast<const void *>(msda::topk_rows_kernel), lds);
    if (e != hipSuccess) return hip_fail(e, "LDS limit of the top-k kernel");
    hipLaunchKernelGGL(msda::topk_rows_kernel, dim3(rows), dim3(msda::kTopkThreads), lds, static_cast<hipStream_t>(stream), scores, n, k,
                       indices, values);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch of the top-k kernel");
    return MSDA_OK;
}

int msda_mask_rows_f32(float *x, const uint8_t *mask, int64_t rows, int row_elems, msda_stream_t stream)
{
    return mask_rows_impl<float>(x, mask, rows, row_elems, stream);
}
int msda_mask_rows_f64(double *x, const uint8_t *mask, int64_t rows, int row_elems, msda_stream_t stream)
{
    return mask_rows_impl<double>(x, mask, rows, row_elems, stream);
}
int msda_mask_rows_bf16(uint16_t *x, const uint8_t *mask, int64_t rows, int row_elems, msda_stream_t stream)
{
    return mask_rows_impl<uint16_t>(x, mask, rows, row_elems, stream);
}

}  // extern "C"
