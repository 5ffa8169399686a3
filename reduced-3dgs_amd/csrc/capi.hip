// capi.hip -- the extern "C" boundary declared in include/r3dgs_rasterizer.h.
//
// Orchestration of one forward / backward pass; replaces CudaRasterizer::Rasterizer::{forward,
// inferenceForward, backward, markVisible} (cuda_rasterizer/rasterizer_impl.cu:149-161, :206-355,
// :359-504, :508-630 of /root/reference/submodules/diff-gaussian-rasterization).
// How the work is issued (results are the same as the reference's):
//   * The reference sizes its binning buffer from a blocking read-back of num_rendered in the middle of every forward
//     (rasterizer_impl.cu:441-450).  Here the hot path (r3dgs_forward_reserved) takes a pair RESERVATION instead:
//     the caller allocates the binning blob for `reserve` pairs up front, every kernel reads the actual count from
//     the device header, and the host never waits.  r3dgs_reserve_hint() proposes the reservation from the
//     num_rendered of earlier passes (published by the passes themselves in host-mapped memory, harvested lazily).
//     r3dgs_forward keeps the reference's exact-size contract (allocator callbacks, returns num_rendered): it waits
//     for that one number by polling host memory -- no HIP call, bounded by a deadline.
//   * All kernels of a pass read their arguments from one device-resident block (common.h), so the launch chain of
//     a shape never changes: it is captured once into a hipGraph and replayed with ONE hipGraphLaunch per pass
//     (+ one hipGraphExecKernelNodeSetParams that carries the new block).  Measured on the MI355X box
//     (tools/launch_bench*.hip): 20 direct launches cost 54 us of host time idle and 0.3-2.3 ms on a loaded /
//     CPU-throttled host, one graph launch 6-10 us either way; a graph with a side-stream branch costs as much as
//     direct launches, so the chain is linear.
//   * no per-call hipMalloc/hipFree: all scratch lives in the three caller blobs;
//   * `debug` and the per-stage timers issue the same chain with direct launches and synchronise / record events
//     between stages (the reference's CHECK_CUDA).
#include "../../include/r3dgs_rasterizer.h"

#include <time.h>

#include <atomic>
#include <random>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace r3 {
int env_int(const char* env, int dflt, int lo, int hi)
{
    const char* v = getenv(env);
    if (!v) return dflt;
    const int p = atoi(v);
    return (p >= lo && p <= hi) ? p : dflt;
}
bool keep_quad_masks()   // R3DGS_KEEP_QUAD_MASKS=0: the backward repeats the forward's region pre-test (A/B runs)
{
    static const bool keep = env_int("R3DGS_KEEP_QUAD_MASKS", 1, 0, 1) != 0;
    return keep;
}
int bwd_segment_log2()
{
    static const int v = env_int("R3DGS_BWD_SEG_LEN", 128, 128, 256) == 256 ? 8 : 7;
    return v;
}
int bwd_segment_factor_pct()
{
    static const int v = env_int("R3DGS_BWD_SEG_FACTOR", 75, 10, 100000);
    return v;
}
int depth_bucket_load()
{
    static const int load = env_int("R3DGS_DEPTH_BUCKET_LOAD", 128, 32, 2048);   // 64 / 128 / 256 / 512 measured: 0.106 / 0.084 / 0.099 / 0.139 ms
    return load;
}
}  // namespace r3

namespace {

using namespace r3;

thread_local std::string g_last_error;
thread_local int g_last_forward_pairs = 0;   // r3dgs_forward_pairs()
thread_local int g_next_forward_trains = 1;  // r3dgs_forward_hint(): holds for this thread's forwards until set again

// Opacity-aware tile rects (gauss_math.h tighten_rect) are the default; R3DGS_TIGHT_RECT=0 / r3dgs_set_tight_rects(0)
// bins into the reference's 3-sigma squares (identical lists to the reference's: the bit-exact binning tests).
std::atomic<int> g_tight_rects{-1};
int tight_rects()
{
    int v = g_tight_rects.load();
    if (v < 0) {
        v = env_int("R3DGS_TIGHT_RECT", 1, 0, 1);
        g_tight_rects.store(v);
    }
    return v;
}

// The backward blend starts its tiles heaviest first (blend.hip unit_order_kernel); R3DGS_TILE_ORDER=0 /
// r3dgs_set_tile_order(0): row-major bands, one per XCD, as the forward (A/B runs, the bit-identity test).
std::atomic<int> g_tile_order{-1};
int heaviest_tiles_first()
{
    int v = g_tile_order.load();
    if (v < 0) {
        v = env_int("R3DGS_TILE_ORDER", 1, 0, 1);
        g_tile_order.store(v);
    }
    return v;
}

// Without a sparsity term the backward takes the SH direction derivatives the forward left (GeomState::sh_ddir) instead of
// Long tile lists are walked by several workgroups of the backward blend, from checkpoints the forward blend leaves
// (common.h, blend.hip); R3DGS_BWD_SEG=0 / r3dgs_set_bwd_segments(0): one workgroup per tile whatever its list
// (A/B runs, the bit-identity tests of the launch order).
std::atomic<int> g_bwd_segments{-1};
int bwd_segments()
{
    int v = g_bwd_segments.load();
    if (v < 0) {
        v = env_int("R3DGS_BWD_SEG", 1, 0, 1);
        g_bwd_segments.store(v);
    }
    return v;
}
// reading every SH row again; R3DGS_SH_CACHE=0 / r3dgs_set_sh_cache(0): it reads the rows (A/B runs, the bit-identity test).
std::atomic<int> g_sh_cache{-1};
int sh_derivative_cache()
{
    int v = g_sh_cache.load();
    if (v < 0) {
        v = env_int("R3DGS_SH_CACHE", 1, 0, 1);
        g_sh_cache.store(v);
    }
    return v;
}

// The per-Gaussian backward evaluates the covariance chain (backward.cu:228-306, 311-374) in double and rounds once
// (gauss_math.h); R3DGS_F64_CHAIN=0 / r3dgs_set_f64_chain(0): the reference's fp32 arithmetic (A/B runs).
std::atomic<int> g_f64_chain{-1};
int f64_chain()
{
    int v = g_f64_chain.load();
    if (v < 0) {
        v = env_int("R3DGS_F64_CHAIN", 1, 0, 1);
        g_f64_chain.store(v);
    }
    return v;
}

bool env_is(const char* name, const char* value)
{
    const char* v = getenv(name);
    return v && std::string(v) == value;
}

size_t cached_depth_temp(size_t P)
{
    static std::mutex mu;
    static std::unordered_map<size_t, size_t> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(P);
    if (it != cache.end()) return it->second;
    size_t b = depth_sort_temp_bytes(P);
    cache[P] = b;
    return b;
}

// ---- the argument-block writers ------------------------------------------------------------------------------------
// The block travels as the kernel's by-value argument (so a replayed graph gets the new block with one
// hipGraphExecKernelNodeSetParams and nothing the host writes later can race with a queued launch) and is copied
// word-wise straight out of the kernarg segment: taking the parameter's address would first spill it to scratch.
template <class Block>
__global__ __launch_bounds__(256) void write_args_kernel(Block* dst, Block v)
{
    static_assert(alignof(Block) == 8 && sizeof(Block) % 4 == 0, "kernarg layout: [dst (8 B)][block]");
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const uint32_t __attribute__((address_space(4))) * KernargWords;
    KernargWords s = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(Block*) / 4;
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    for (uint32_t k = threadIdx.x; k < sizeof(Block) / 4; k += 256) d[k] = s[k];
#endif
    (void)v;
}

// ---- optional per-stage timing with HIP events on the caller's stream (r3dgs_profile_*) ----------
struct Profiler {
    std::mutex mu;
    std::atomic<unsigned> mask{0};   // bit (stage): record an event pair around that stage
    std::vector<std::pair<hipEvent_t, hipEvent_t>> used[kNumStages];
    std::vector<hipEvent_t> pool;
    hipEvent_t get()
    {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        R3_HIP(hipEventCreate(&e));
        return e;
    }
} g_prof;
constexpr unsigned kFwdStages = (1u << kPre) | (1u << kDepthSort) | (1u << kBinning) | (1u << kBlendFwd) | (1u << kColor);
constexpr unsigned kBwdStages = (1u << kBlendBwd) | (1u << kPreBwd) | (1u << kBlendBwdKernel);

struct StageRun {   // direct-issue mode: events and / or debug synchronisation around each stage
    bool debug;
    hipEvent_t a[kNumStages] = {};
    void begin(int stage, hipStream_t s)
    {
        if (!((g_prof.mask.load() >> stage) & 1u)) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        a[stage] = g_prof.get();
        R3_HIP(hipEventRecord(a[stage], s));
    }
    void end(int stage, const char* what, hipStream_t s)
    {
        if (a[stage]) {
            std::lock_guard<std::mutex> lk(g_prof.mu);
            hipEvent_t b = g_prof.get();
            R3_HIP(hipEventRecord(b, s));
            g_prof.used[stage].push_back({a[stage], b});
            a[stage] = nullptr;
        }
        check_launch(what, s, debug);
    }
};

// ---- pass tickets and the host-mapped PassInfo ring ---------------------------------------------------------------
constexpr uint32_t kInfoRing = 1024;
struct InfoRing {
    PassInfo* host = nullptr;
    PassInfo* dev = nullptr;
};
std::mutex g_ring_mu;
std::unordered_map<int, InfoRing> g_rings;
std::atomic<uint64_t> g_next_ticket{1};
// A ticket = (device << 48) | running pass number: whoever holds one can find its ring without knowing which device is
// current (r3dgs_pass_query from another thread / with another GPU selected).
constexpr int kTicketDevShift = 48;
inline uint64_t ticket_number(uint64_t t) { return t & ((1ull << kTicketDevShift) - 1ull); }
inline int ticket_device(uint64_t t) { return (int)(t >> kTicketDevShift); }

InfoRing& info_ring(int dev)
{
    std::lock_guard<std::mutex> lk(g_ring_mu);
    InfoRing& r = g_rings[dev];
    if (!r.host) {
        R3_HIP(hipHostMalloc(reinterpret_cast<void**>(&r.host), sizeof(PassInfo) * kInfoRing,
                             hipHostMallocMapped | hipHostMallocCoherent));
        memset(r.host, 0, sizeof(PassInfo) * kInfoRing);
        R3_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&r.dev), r.host, 0));
    }
    return r;
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Wait until the pass has published its header (num_rendered, visible).  Plain loads of host memory: a spin of up to
// ~100 us (R3DGS_SPIN_US), then sleeps -- a spinning thread burns CPU quota that a containerised trainer may not have,
// but a sleep of "5 us" returns after 50-60 (timer slack), which a 10 k-Gaussian scene whose whole step takes 0.15 ms of
// GPU time pays on every forward (4000 instead of 5600 it/s) -- and a deadline (R3DGS_SYNC_TIMEOUT_MS, default 30 s) that
// turns a hung GPU into an error instead of a hung host.
const volatile PassInfo* wait_info(uint64_t ticket)
{
    static const int timeout_ms = env_int("R3DGS_SYNC_TIMEOUT_MS", 30000, 1, 3600000);
    static const int spin_us = env_int("R3DGS_SPIN_US", 100, 0, 1000000);
    const volatile PassInfo* info = info_ring(ticket_device(ticket)).host + ticket_number(ticket) % kInfoRing;
    const uint32_t seq = (uint32_t)ticket_number(ticket);
    const double t0 = now_ms();
    for (int spins = 0;; spins++) {
        if (info->seq == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return info;
        }
        if ((spins & 63) != 63) continue;   // look at the clock every 64 polls
        const double waited = now_ms() - t0;
        if (waited * 1000.0 < (double)spin_us) continue;
        if (waited > timeout_ms)
            throw Error("timed out after " + std::to_string(timeout_ms) + " ms waiting for num_rendered of pass " +
                        std::to_string(ticket_number(ticket)) + " (GPU hung or stream never ran?)");
        timespec ts = {0, waited < 2.0 ? 5000 : 50000};
        nanosleep(&ts, nullptr);
    }
}

// ---- reservation advice from the passes seen so far ----------------------------------------------------------------
// The reservation of a pass is a guess; r3dgs_pass_query says whether it held (R3DGS_PASS_TRUNCATED), and the host side
// redoes a truncated pass on the exact-size path before anything consumes it (diff_gaussian_rasterization/_C.py), so the
// advice only decides how often that happens.  Pair counts differ by 2-3x between the cameras of a real scene, so they
// are remembered per CAMERA (key: the device address of its view matrix -- every scene/cameras.py Camera owns one),
// with the per-image-size maximum as the answer for a camera not seen yet.
struct Pending {
    uint64_t ticket;
    int dev, P, W, H;
    uint32_t reserve;   // 0xFFFFFFFF for exact-size passes
    uintptr_t cam;
    bool sort_stamp;    // the pass's depth scan stamps PassInfo::sort_seq (bucketed sort)
};
struct CamStat {
    int P;
    uint32_t pairs;
};
struct ViewStats {
    std::deque<std::pair<int, uint32_t>> recent;   // (P, pairs) of the last passes of this (device, W, H)
    std::unordered_map<uintptr_t, CamStat> cams;   // last pass of each camera
    int prefer_generic = 0;                        // passes left to route through the generic depth sort
};
struct Advisor {
    std::mutex mu;
    std::vector<Pending> pending;
    std::map<std::tuple<int, int, int>, ViewStats> views;
    uint64_t overflow_events = 0;
    uint32_t last_overflow_rendered = 0, last_overflow_reserve = 0;
} g_adv;
constexpr size_t kRecentWindow = 1024;
constexpr size_t kMaxCams = 16384;

void observe_locked(const Pending& p, uint32_t pairs, uint32_t sort_overflow)
{
    ViewStats& v = g_adv.views[{p.dev, p.W, p.H}];
    v.recent.push_back({p.P, pairs});
    if (v.recent.size() > kRecentWindow) v.recent.pop_front();
    if (p.cam) {
        if (v.cams.size() > kMaxCams) v.cams.clear();
        v.cams[p.cam] = {p.P, pairs};
    }
    if (sort_overflow) v.prefer_generic = 64;
    if (p.reserve != 0xFFFFFFFFu && pairs > p.reserve) {
        g_adv.overflow_events++;
        g_adv.last_overflow_rendered = pairs;
        g_adv.last_overflow_reserve = p.reserve;
    }
}

// Passes that have published their numbers since the last call.  Entries are looked at independently: a pass sitting on
// a blocked stream (or one whose launch failed) does not hold up the ones behind it.
void harvest_locked()
{
    const uint64_t newest = g_next_ticket.load();
    size_t keep = 0;
    for (size_t k = 0; k < g_adv.pending.size(); k++) {
        const Pending p = g_adv.pending[k];
        const uint64_t n = ticket_number(p.ticket);
        if (newest - n >= kInfoRing - 8) continue;   // its slot is about to be / was reused: forget it
        const volatile PassInfo* info = info_ring(p.dev).host + n % kInfoRing;
        // the overflow hint of the bucketed depth sort comes from a later kernel than the header: its own stamp
        const bool there = info->seq == (uint32_t)n && (!p.sort_stamp || info->sort_seq == (uint32_t)n);
        if (!there) {
            g_adv.pending[keep++] = p;
            continue;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        observe_locked(p, info->pairs, p.sort_stamp ? info->sort_overflow : 0u);
    }
    g_adv.pending.resize(keep);
    if (g_adv.pending.size() > 4 * kInfoRing) g_adv.pending.clear();
}

void forget_ticket(uint64_t ticket)   // the launch behind a ticket failed: nothing will ever publish it
{
    std::lock_guard<std::mutex> lk(g_adv.mu);
    for (size_t k = 0; k < g_adv.pending.size(); k++)
        if (g_adv.pending[k].ticket == ticket) {
            g_adv.pending.erase(g_adv.pending.begin() + (long)k);
            return;
        }
}

// Reservations come from a geometric grid (steps of 2^(1/8) ~ 9 %): a reservation keys the captured graph of its shape,
// and per-camera pair counts would otherwise make every camera a shape of its own.
uint32_t quantize_reserve(double want)
{
    if (want >= 2147000000.0) return 0;   // beyond a 31-bit pair count: exact path decides
    const double unit = 65536.0;
    int k = want <= unit ? 0 : (int)ceil(8.0 * log2(want / unit) - 1e-9);
    double v = unit * exp2((double)k / 8.0);
    while (v < want) v = unit * exp2((double)(++k) / 8.0);
    const uint64_t r = ((uint64_t)v + 4095u) / 4096u * 4096u;
    return r >= 2147000000ull ? 0u : (uint32_t)r;
}

uint32_t reserve_hint(int dev, int P, int W, int H, uintptr_t cam)
{
    static const bool off = env_is("R3DGS_RESERVE", "off");
    static const double slack = 0.01 * env_int("R3DGS_RESERVE_SLACK_PCT", 150, 100, 1600);
    if (off || P <= 0) return 0;
    std::lock_guard<std::mutex> lk(g_adv.mu);
    harvest_locked();
    auto it = g_adv.views.find({dev, W, H});
    if (it == g_adv.views.end() || it->second.recent.empty()) return 0;
    ViewStats& v = it->second;
    // scaled to the current Gaussian count (densification / pruning between passes)
    auto scaled = [P](int p0, uint32_t pairs) { return p0 > 0 ? (double)pairs * ((double)P / (double)p0) : (double)pairs; };
    double base = 0.0;
    auto c = cam ? v.cams.find(cam) : v.cams.end();
    double recent_max = 0.0;
    for (auto& e : v.recent) recent_max = std::max(recent_max, scaled(e.first, e.second));
    if (c != v.cams.end()) {
        // never far below what this image size has needed lately: a camera entry can be stale (a caller that reuses one
        // device buffer for every camera's matrix; an address the allocator handed to another camera -- the host layer
        // reports freed matrices through r3dgs_reserve_forget_view, a C caller may not)
        base = std::max(scaled(c->second.P, c->second.pairs), 0.3 * recent_max);
    } else {
        base = recent_max;
    }
    return quantize_reserve(base * slack + 65536.0);
}

// ---- launch contexts: device argument blocks and captured graphs ---------------------------------------------------
struct CtxKey {
    int dev, kind;   // kind 0: forward, 1: backward
    hipStream_t stream;
    int P, M, W, H;
    uint32_t reserve, flags;
    bool operator<(const CtxKey& o) const
    {
        return std::tie(dev, kind, stream, P, M, W, H, reserve, flags) <
               std::tie(o.dev, o.kind, o.stream, o.P, o.M, o.W, o.H, o.reserve, o.flags);
    }
};
struct GraphCtx {
    void* d_args = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraphNode_t node0 = nullptr;
    hipKernelNodeParams node0_params = {};
    uint64_t last_use = 0;
};
std::mutex g_ctx_mu;
std::map<CtxKey, GraphCtx> g_graphs;
std::map<std::tuple<int, int, hipStream_t>, void*> g_direct_blocks;   // (device, kind, stream) -> device block
std::unordered_map<int, hipStream_t> g_capture_streams;
uint64_t g_use_clock = 0;
constexpr size_t kMaxGraphs = 48;

void* direct_block(int dev, int kind, hipStream_t s, size_t bytes)
{
    void*& p = g_direct_blocks[{dev, kind, s}];
    if (!p) R3_HIP(hipMalloc(&p, bytes));
    return p;
}

hipStream_t capture_stream(int dev)
{
    hipStream_t& s = g_capture_streams[dev];
    if (!s) R3_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}

void evict_graphs_locked()
{
    while (g_graphs.size() > kMaxGraphs) {
        auto victim = g_graphs.begin();
        for (auto it = g_graphs.begin(); it != g_graphs.end(); ++it)
            if (it->second.last_use < victim->second.last_use) victim = it;
        // the exec may still be queued on its stream: wait for that stream before tearing it down (rare path)
        (void)hipStreamSynchronize(victim->first.stream);
        if (victim->second.exec) (void)hipGraphExecDestroy(victim->second.exec);
        if (victim->second.graph) (void)hipGraphDestroy(victim->second.graph);
        if (victim->second.d_args) (void)hipFree(victim->second.d_args);
        g_graphs.erase(victim);
    }
}

// Second stream of a forward whose colour stream runs BESIDE the depth sort and the binning instead of inside the sort's
// launches (FwdPlan::color_side, large scenes): forked after the geometry kernel, joined before the blend.  One per device;
// the fork / join events carry no timing and may be re-recorded while an earlier wait is pending (a wait refers to the
// record that preceded it).  Inside a stream capture the two event waits make the side stream part of the captured graph.
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
std::mutex g_side_mu;
std::unordered_map<int, SideStream> g_side;
SideStream& side_stream()
{
    int dev = 0;
    R3_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_side_mu);
    SideStream& ss = g_side[dev];
    if (!ss.stream) {
        R3_HIP(hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking));
        R3_HIP(hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming));
        R3_HIP(hipEventCreateWithFlags(&ss.join, hipEventDisableTiming));
    }
    return ss;
}

bool graphs_enabled()
{
    static const bool on = !env_is("R3DGS_GRAPH", "0");
    return on;
}

// ---- the launch chains -----------------------------------------------------------------------------------------------
struct NoHooks {
    void begin(int, hipStream_t) {}
    void end(int, const char*, hipStream_t) {}
};

// phases: 1 = geometry (+ header: everything num_rendered depends on), 2 = the rest.  The geometry kernel installs the
// pass block `args` at `d` for the kernels behind it; a caller that issues phase 2 on its own (exact-size path) writes
// the completed block with write_args first.
template <class Hooks>
void issue_forward(const FwdPlan& p, FwdPassArgs* d, const FwdPassArgs& args, hipStream_t s, int phases, Hooks& h,
                   GeomState* g_host)
{
    if (phases & 1) {
        h.begin(kPre, s);
        issue_preprocess_geom(p, d, args, s);
        h.end(kPre, "preprocess", s);
        if (!args.depth.fuse_header) issue_header_reduce(&d->header, s);
    }
    if (phases & 2) {
        SideStream* side = nullptr;
        // (one fork / join event pair per device: two host threads issuing forwards at once would interleave record and
        // wait of each other's passes, so the issue of a pass that uses the side stream is serialised -- opt-in A/B path)
        static std::mutex side_issue_mu;
        std::unique_lock<std::mutex> side_lk(side_issue_mu, std::defer_lock);
        if (p.color_side) {   // the colour stream on a stream of its own, beside the sort and the binning
            side_lk.lock();
            side = &side_stream();
            R3_HIP(hipEventRecord(side->fork, s));
            R3_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
            issue_preprocess_color(p, &d->pre, side->stream);
            R3_HIP(hipEventRecord(side->join, side->stream));
        }
        h.begin(kDepthSort, s);
        if (p.generic_depth_sort)
            run_generic_depth_sort(p.P, *g_host, s);
        else
            issue_depth_sort_and_color(p, d, s);   // the SH -> RGB stream rides in spare workgroups of these kernels
        h.end(kDepthSort, "depth sort + scan", s);
        if (p.generic_depth_sort && !p.color_in_geom && !p.color_side) {
            h.begin(kColor, s);
            issue_preprocess_color(p, &d->pre, s);
            h.end(kColor, "SH colours", s);
        }
        h.begin(kBinning, s);
        issue_tile_binning(p, d, s);
        h.end(kBinning, "tile binning", s);
        if (side) R3_HIP(hipStreamWaitEvent(s, side->join, 0));
        h.begin(kBlendFwd, s);
        issue_blend_forward(p, &d->blend, s);
        h.end(kBlendFwd, "blend forward", s);
    }
}

template <class Block>
void launch_write_args(Block* dst, const Block& v, hipStream_t s);

// The backward blend installs the pass block; without pairs (empty reservation) it does not run and write_args does.
template <class Hooks>
void issue_backward(const BwdPlan& p, BwdPassArgs* d, const BwdPassArgs& args, hipStream_t s, Hooks& h)
{
    h.begin(kBlendBwd, s);
    // the pair flags are all zero here: the forward's tile_ranges kernel clears them and pair_reduce puts every
    // flag it consumed back to zero, so neither pass pays for a fill of its own
    if (p.has_pairs) {
        issue_unit_order(p, d, args, s);
        h.begin(kBlendBwdKernel, s);
        issue_blend_backward(p, d, args, s);
        h.end(kBlendBwdKernel, "blend backward kernel", s);
        issue_pair_reduce(p, &d->reduce, s);
    } else {
        launch_write_args(d, args, s);
    }
    h.end(kBlendBwd, "blend backward", s);
    h.begin(kPreBwd, s);
    issue_preprocess_backward(p, &d->pre, s);
    h.end(kPreBwd, "preprocess backward", s);
}

template <class Block>
void launch_write_args(Block* dst, const Block& v, hipStream_t s)
{
    hipLaunchKernelGGL(write_args_kernel<Block>, dim3(1), dim3(256), 0, s, dst, v);
}

// Replays (capturing it first if this shape is new) the launch chain of a pass as one graph launch on `s`.  Node 0 of
// the chain is the kernel that receives (Block* dst, Block by value) and installs the block; on replay only its
// arguments are refreshed (hipGraphExecKernelNodeSetParams copies them at the call, so a host running many passes
// ahead cannot disturb launches that are still queued -- checked in tools/launch_bench2.hip with the GPU kept busy).
template <class Block, class Plan, class IssueFn>
void launch_graph(const CtxKey& key, const Plan& plan, const Block& args, hipStream_t s, IssueFn&& issue)
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    GraphCtx& c = g_graphs[key];
    if (!c.exec) {
        try {
            R3_HIP(hipMalloc(&c.d_args, sizeof(Block)));
            hipStream_t cap = capture_stream(key.dev);
            R3_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
            try {
                issue(plan, static_cast<Block*>(c.d_args), args, cap);
            } catch (...) {
                hipGraph_t dead = nullptr;
                (void)hipStreamEndCapture(cap, &dead);
                if (dead) (void)hipGraphDestroy(dead);
                throw;
            }
            R3_HIP(hipStreamEndCapture(cap, &c.graph));
            size_t n_root = 0;
            R3_HIP(hipGraphGetRootNodes(c.graph, nullptr, &n_root));
            if (n_root != 1) throw Error("captured pass graph is not a linear chain");
            R3_HIP(hipGraphGetRootNodes(c.graph, &c.node0, &n_root));
            R3_HIP(hipGraphKernelNodeGetParams(c.node0, &c.node0_params));
            R3_HIP(hipGraphInstantiate(&c.exec, c.graph, nullptr, nullptr, 0));
            c.last_use = ++g_use_clock;   // before the eviction below: the newest entry must not be its victim
        } catch (...) {
            if (c.graph) (void)hipGraphDestroy(c.graph);
            if (c.d_args) (void)hipFree(c.d_args);
            g_graphs.erase(key);
            throw;
        }
        evict_graphs_locked();
    }
    GraphCtx& ctx = g_graphs[key];
    ctx.last_use = ++g_use_clock;
    Block* dst = static_cast<Block*>(ctx.d_args);
    Block copy = args;
    void* params[2] = {&dst, &copy};
    hipKernelNodeParams kp = ctx.node0_params;   // function, grid, block as captured
    kp.kernelParams = params;
    kp.extra = nullptr;
    R3_HIP(hipGraphExecKernelNodeSetParams(ctx.exec, ctx.node0, &kp));   // by-value block: copied at this call
    R3_HIP(hipGraphLaunch(ctx.exec, s));
}

// ---- argument blocks -------------------------------------------------------------------------------------------------
struct FwdCall {
    int P;
    const int* D;
    int M;
    const int *coeffsNum, *perBand, *cumSum;
    const float* background;
    int width, height;
    const float *means3D, *shs, *colors_precomp, *opacities, *scales;
    float scale_modifier;
    const float *rotations, *cov3D_precomp, *viewmatrix, *projmatrix, *cam_pos;
    float tan_fovx, tan_fovy;
    float* out_color;
    int* out_touched_pixels;
    float* out_transmittance;
    int* radii;
    int calculate_mean_transmittance, debug;
    hipStream_t stream;
};

void validate_forward(const FwdCall& c)
{
    if (!c.means3D || !c.opacities || !c.viewmatrix || !c.projmatrix || !c.cam_pos || !c.background || !c.out_color)
        throw Error("a required pointer is NULL");
    if (!c.colors_precomp && !c.shs) throw Error("provide SHs or precomputed colours");
    if (!c.cov3D_precomp && (!c.scales || !c.rotations)) throw Error("provide scale/rotation or a precomputed 3D covariance");
    if (!c.colors_precomp && !c.coeffsNum && (c.M < 1 || c.M > 16)) throw Error("SH coefficient count M must be in [1,16]");
    if (!c.colors_precomp && !c.coeffsNum && !c.D) throw Error("per-Gaussian degrees must be provided with SHs");
    if (c.width <= 0 || c.height <= 0) throw Error("image size must be positive");
    if (c.calculate_mean_transmittance && (!c.out_touched_pixels || !c.out_transmittance))
        throw Error("counter mode needs out_touched_pixels and out_transmittance");
}

FwdPlan make_fwd_plan(const FwdCall& c, uint32_t reserve)
{
    static const int ppl0 = env_int("R3DGS_FWD_PPL", 1, 1, 4);   // 1 / 2 / 4 measured: 0.179 / 0.188 / 0.225 ms
    // workgroups of the colour stream per launch that carries it: P / 512 between 512 and 4096 (measured: 1024 vs 512 at
    // 500 k: stage 0.086 vs 0.088 ms; 4096 vs 512 at 2 M: 0.259 vs 0.299 ms); R3DGS_COLOR_GRID overrides
    static const int color_grid_env = env_int("R3DGS_COLOR_GRID", -1, 0, 1 << 20);
    static const bool generic_env = env_is("R3DGS_DEPTH_SORT", "generic");   // forces the rocPRIM path (A/B runs, tests)
    FwdPlan p;
    p.P = c.P;
    p.M = c.M;
    p.W = c.width;
    p.H = c.height;
    p.gx = (c.width + kTile - 1) / kTile;
    p.gy = (c.height + kTile - 1) / kTile;
    p.reserve = reserve;
    p.grid_pairs = grid_pairs_for(reserve);
    p.layout = pair_layout(c.P, (size_t)p.gx * p.gy);
    p.nb = depth_bucket_count((size_t)c.P);
    p.ragged = (c.coeffsNum != nullptr && !c.colors_precomp) ? 1 : 0;
    p.counters = c.calculate_mean_transmittance ? 1 : 0;
    p.fwd_ppl = ppl0 == 3 ? 2 : ppl0;
    p.color_grid = color_grid_env >= 0 ? color_grid_env : std::min(4096, std::max(512, c.P / 512));
    static const int fuse = env_int("R3DGS_COLOR_FUSE", 1, 0, 1);
    // share of the colour chunks carried by the histogram / scatter / bucket-sort launches: 25 / 45 / 30 % up to 1 M Gaussians
    // (round 6, profiles/r06_sweep_colour_split.txt: depth sort + colour 0.0896 against 0.0929 ms with round 5's 20 / 35 / 45 at
    // 500 k, both alternating rounds), 25 / 25 / 50 % above (2 M: stage 0.257 vs 0.266 ms, 6 M: 0.762 vs 0.778)
    static const int split0_env = env_int("R3DGS_COLOR_SPLIT0", -1, 0, 100), split1_env = env_int("R3DGS_COLOR_SPLIT1", -1, 0, 100);
    const bool big = c.P > (1 << 20);
    const int split0 = split0_env >= 0 ? split0_env : 25, split1 = split1_env >= 0 ? split1_env : big ? 25 : 45;
    p.color_fuse = fuse;
    p.color_split[0] = split0;
    p.color_split[1] = split0 + split1 > 100 ? 100 - split0 : split1;
    p.color_split[2] = 100 - p.color_split[0] - p.color_split[1];
    p.generic_depth_sort = (generic_env || c.P >= (1 << 24)) ? 1 : 0;   // the bucket histogram packs the count in 24 bits
    static const int in_geom_env = env_int("R3DGS_COLOR_IN_GEOM", -1, -1, 1);
    p.color_in_geom = in_geom_env > 0 ? 1 : 0;
    if (p.color_in_geom) p.color_fuse = 0;
    static const int side_env = env_int("R3DGS_COLOR_STREAM", -1, -1, 1);
    p.color_side = (side_env > 0 && !p.color_in_geom) ? 1 : 0;
    if (p.color_side) p.color_fuse = 0;
    p.tight = tight_rects();
    return p;
}

void fill_fwd_args(FwdPassArgs& a, const FwdPlan& p, const FwdCall& c, const GeomState& g, const BinState* b,
                   const ImageState& img, PassInfo* info_dev, uint64_t ticket, bool fuse_header)
{
    memset(&a, 0, sizeof(a));
    int* radii = c.radii ? c.radii : g.radii_internal;
    FwdInputs& in = a.pre.in;
    in.P = c.P;
    in.M = c.M;
    in.degrees = c.D;
    in.means3D = c.means3D;
    in.scales = c.scales;
    in.rotations = c.rotations;
    in.opacities = c.opacities;
    in.shs = c.shs;
    in.cov3D_precomp = c.cov3D_precomp;
    in.colors_precomp = c.colors_precomp;
    in.coeffs_num = p.ragged ? c.coeffsNum : nullptr;
    in.per_band_count = p.ragged ? c.perBand : nullptr;
    in.cumsum_count = p.ragged ? c.cumSum : nullptr;
    ViewParams& v = a.pre.view;
    v.view = c.viewmatrix;
    v.proj = c.projmatrix;
    v.campos = c.cam_pos;
    v.bg = c.background;
    v.tan_fovx = c.tan_fovx;
    v.tan_fovy = c.tan_fovy;
    v.W = c.width;
    v.H = c.height;
    v.scale_modifier = c.scale_modifier;
    a.pre.rec = g.rec;
    a.pre.rect = g.rect;
    a.pre.depth_key = g.depth_key;
    a.pre.tiles = g.tiles;
    a.pre.partials = g.partials;
    a.pre.radii = radii;
    a.pre.color_blocks = (c.P + kPreBlockSize - 1) / kPreBlockSize;
    a.pre.tight = p.tight;
    // (a forward that its caller knows no backward will follow -- r3dgs_forward_hint(0) -- leaves nothing; the header says so)
    a.pre.sh_ddir = (g_next_forward_trains && !p.ragged && c.shs && !c.colors_precomp) ? g.sh_ddir : nullptr;

    a.header.parts = g.partials;
    a.header.n_parts = (int)pre_partials((size_t)c.P);
    a.header.hdr = g.header;
    a.header.info = info_dev;
    a.header.ticket = (uint32_t)ticket_number(ticket);
    a.header.reserve = p.reserve;
    a.header.stamp_sort = p.generic_depth_sort ? 1 : 0;   // no scan kernel behind the header on the generic-sort route
    a.header.sh_cache = a.pre.sh_ddir ? 1u : 0u;
    // checkpoints for a split backward walk: only a forward that a backward will follow leaves them
    // (the header is written in the exact-size path's first phase, before the binning blob exists: what it says must not
    // depend on `b`)
    const bool leave_ckpt = g_next_forward_trains && bwd_segments() && !p.counters;
    float4* const ckpt = (b && leave_ckpt) ? b->ckpt : nullptr;
    a.header.ckpt = leave_ckpt ? (uint32_t)bwd_segment_log2() : 0u;
    a.header.ckpt_factor_pct = (uint32_t)bwd_segment_factor_pct();
    a.header.n_tiles = (uint32_t)(p.gx * p.gy);

    DepthArgs& d = a.depth;
    d.P = c.P;
    d.nb = p.nb;
    d.rows = (int)depth_hist_rows((size_t)c.P);
    d.per_block = (int)depth_hist_per_block((size_t)c.P);
    d.fuse_header = fuse_header ? 1 : 0;
    d.key = g.depth_key;
    d.tiles = g.tiles;
    d.hdr = g.header;
    d.info = info_dev;
    d.ticket = (uint32_t)ticket_number(ticket);
    d.ds = g.dsort;
    d.hist_rows = g.hist_rows;
    d.hist_base = g.hist_base;
    d.key_sorted = g.key_sorted;
    d.bucket_id = g.bucket_id;
    d.ovf_key = g.ovf_key;
    d.ovf_id = g.ovf_id;
    d.rect = g.rect;
    d.rec16 = g.rec16;
    d.rec16_b = g.rec16_b;
    d.rect_sorted = g.rect_sorted;
    d.order = g.order;
    d.offsets = g.offsets;
    // only on the asynchronous path does the binning blob exist while the depth sort runs (and only the bucketed sort's
    // scan writes the notes)
    uint32_t* const block_first = (b && fuse_header && !p.generic_depth_sort) ? b->block_first : nullptr;
    d.block_first = block_first;
    d.block_cap = p.reserve / (uint32_t)kRadixBlock + 2u;

    const PairLayout& l = p.layout;
    const size_t Tn = (size_t)p.gx * p.gy;
    if (b) {
        const uint32_t stride = radix_row_stride(p.reserve);
        EmitArgs& e = a.emit;
        e.P = c.P;
        e.gx = p.gx;
        e.hdr = g.header;
        e.order = g.order;
        e.offsets = g.offsets;
        e.block_first = block_first;
        e.rect = g.rect;
        e.rect_sorted = p.generic_depth_sort ? nullptr : g.rect_sorted;
        e.rec = g.rec;
        e.rank_bits = l.rank_bits;
        e.digit_bits = l.digit_bits;
        e.words_out = b->words_a;
        e.cap = p.reserve;
        e.pair_rank = l.wide == 1 ? b->pair_rank : nullptr;
        e.ranges = img.ranges;
        e.n_tiles = (uint32_t)Tn;
        e.radix_rows = b->radix_rows;
        e.row_stride = stride;
        // buffer chain of the passes: narrow a -> b -> c (-> b), wide a -> b -> a (-> b); words_a of the narrow layout
        // survives for the backward (pair_rank aliases it)
        const char* src = b->words_a;
        for (int k = 0; k < l.passes; k++) {
            char* dst = l.wide ? ((k & 1) ? b->words_a : b->words_b) : ((k & 1) ? b->words_c : b->words_b);
            RadixArgs& r = a.radix[k];
            r.hdr = g.header;
            r.in = src;
            r.out = dst;
            r.ids_out = (l.wide == 2 && k == l.passes - 1) ? b->point_list : nullptr;
            r.cap = p.reserve;
            r.shift = l.rank_bits + k * l.digit_bits;
            r.digit_bits = l.digit_bits;
            r.rank_bits = l.rank_bits;
            r.tile_shift = k * l.digit_bits;
            r.rows = b->radix_rows;
            r.base = b->radix_base;
            r.total = b->radix_total + k * kMaxRadixBins;
            r.row_stride = stride;
            src = dst;
        }
        if (src != sorted_words(*b, l)) throw Error("internal: sorted-word buffer mismatch");
        RangesArgs& t = a.ranges;
        t.hdr = g.header;
        t.sorted = src;
        t.cap = p.reserve;
        t.rank_bits = l.rank_bits;
        t.order = nullptr;   // the words carry Gaussian ids
        t.point_list = b->point_list;
        t.ranges = img.ranges;
        t.pair_flag = b->pair_flag;
    }
    BlendFwdArgs& f = a.blend;
    f.ranges = img.ranges;
    f.point_list = b ? b->point_list : nullptr;
    f.rec = g.rec;
    f.W = c.width;
    f.H = c.height;
    f.gx = p.gx;
    f.nblocks = (uint32_t)(p.gx * p.gy * (4 / p.fwd_ppl));
    f.bg = c.background;
    f.out_color = c.out_color;
    f.final_T = img.final_T;
    f.n_contrib = img.n_contrib;
    f.touched = p.counters ? c.out_touched_pixels : nullptr;
    f.transmittance = p.counters ? c.out_transmittance : nullptr;
    f.quad_masks = b && keep_quad_masks() ? b->quad_masks : nullptr;
    f.quad_depth = img.quad_depth;
    f.ckpt = ckpt;
    f.hdr = g.header;
}

uint32_t fwd_flags(const FwdPlan& p, const FwdCall& c)
{
    return (uint32_t)p.ragged | ((uint32_t)p.counters << 1) | ((uint32_t)p.fwd_ppl << 2) |
           ((uint32_t)p.layout.wide << 8) | ((uint32_t)(c.colors_precomp != nullptr) << 6) | ((uint32_t)p.color_fuse << 7) |
           ((uint32_t)p.tight << 10) | ((uint32_t)p.color_in_geom << 11) | ((uint32_t)p.color_side << 12);
}

int current_device()
{
    int dev = 0;
    R3_HIP(hipGetDevice(&dev));
    return dev;
}

uint64_t new_ticket(int dev, const FwdCall& c, uint32_t reserve, bool sort_stamp, PassInfo** info_dev)
{
    const uint64_t number = g_next_ticket.fetch_add(1);
    const uint64_t ticket = ((uint64_t)dev << kTicketDevShift) | number;
    InfoRing& ring = info_ring(dev);
    *info_dev = ring.dev + number % kInfoRing;
    std::lock_guard<std::mutex> lk(g_adv.mu);
    g_adv.pending.push_back({ticket, dev, c.P, c.width, c.height, reserve, reinterpret_cast<uintptr_t>(c.viewmatrix), sort_stamp});
    return ticket;
}

bool take_prefer_generic(int dev, const FwdCall& c)
{
    std::lock_guard<std::mutex> lk(g_adv.mu);
    auto it = g_adv.views.find({dev, c.width, c.height});
    if (it == g_adv.views.end() || it->second.prefer_generic <= 0) return false;
    it->second.prefer_generic--;
    return true;
}

size_t geometry_blob_bytes(size_t P, size_t depth_temp, bool lean)
{
    const GeomState g = GeomState::carve(nullptr, P, depth_temp);
    return (size_t)reinterpret_cast<uintptr_t>(lean ? g.lean_end : g.end) + kAlign;
}

// Exact-size forward (the reference's contract): allocator callbacks, returns num_rendered.
int forward_exact(r3dgs_alloc_fn geometryBuffer, void* geometry_user, r3dgs_alloc_fn binningBuffer, void* binning_user,
                  r3dgs_alloc_fn imageBuffer, void* image_user, const FwdCall& c)
{
    g_last_forward_pairs = 0;
    if (c.P <= 0) return 0;
    if (!geometryBuffer || !binningBuffer || !imageBuffer) throw Error("allocator callbacks must not be NULL");
    validate_forward(c);
    hipStream_t s = c.stream;
    const int dev = current_device();
    FwdPlan plan = make_fwd_plan(c, 0xFFFFFFFFu);
    if (!plan.generic_depth_sort && take_prefer_generic(dev, c)) plan.generic_depth_sort = 1;
    prepare_depth_bucket_sort(plan.nb);
    const size_t depth_temp = cached_depth_temp((size_t)c.P);
    const bool lean = !(g_next_forward_trains && !plan.ragged && c.shs && !c.colors_precomp);   // no sh_ddir will be written
    char* gptr = geometryBuffer(geometry_blob_bytes((size_t)c.P, depth_temp, lean), geometry_user);
    if (!gptr) throw Error("geometry allocator returned NULL");
    GeomState geom = GeomState::carve(gptr, (size_t)c.P, depth_temp);
    const size_t N = (size_t)c.width * c.height, Tn = (size_t)plan.gx * plan.gy;
    char* iptr = imageBuffer(required_bytes<ImageState>(N, Tn), image_user);
    if (!iptr) throw Error("image allocator returned NULL");
    ImageState img = ImageState::carve(iptr, N, Tn);

    PassInfo* info_dev = nullptr;
    const uint64_t ticket = new_ticket(dev, c, 0xFFFFFFFFu, false, &info_dev);
    FwdPassArgs* d_args;
    StageRun hooks{c.debug != 0};
    FwdPassArgs args;
    try {
        {
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            d_args = static_cast<FwdPassArgs*>(direct_block(dev, 0, s, sizeof(FwdPassArgs)));
        }
        fill_fwd_args(args, plan, c, geom, nullptr, img, info_dev, ticket, false);
        issue_forward(plan, d_args, args, s, 1, hooks, &geom);
    } catch (...) {
        forget_ticket(ticket);
        throw;
    }
    // the one structural wait of this entry point (the reference's cudaMemcpy at rasterizer_impl.cu:446)
    const volatile PassInfo* info = wait_info(ticket);
    const uint32_t R = info->pairs, R_ref = info->num_rendered;
    if (R > 0x7fffffffu || R_ref > 0x7fffffffu) throw Error("num_rendered exceeds 2^31-1");
    g_last_forward_pairs = (int)R;
    // The blob is carved for the count this entry point RETURNS (the reference's num_rendered, >= the pairs binned with the
    // opacity-aware rects): a caller that follows the reference's contract hands that number to r3dgs_backward /
    // r3dgs_export_binning as R, and both carve the blob with it.  Grids are sized by the pairs.
    plan.reserve = R_ref ? R_ref : 1u;
    plan.grid_pairs = R ? R : 1u;     // exact size: one block per chunk
    char* bptr = binningBuffer(required_bytes<BinState>((size_t)plan.reserve, plan.layout.wide, TileGrid{(uint32_t)plan.gx, (uint32_t)plan.gy}), binning_user);
    if (!bptr) throw Error("binning allocator returned NULL");
    BinState bin = BinState::carve(bptr, (size_t)plan.reserve, plan.layout.wide, TileGrid{(uint32_t)plan.gx, (uint32_t)plan.gy});
    fill_fwd_args(args, plan, c, geom, &bin, img, info_dev, ticket, false);
    launch_write_args(d_args, args, s);   // the completed block (binning pointers, pair capacity)
    issue_forward(plan, d_args, args, s, 2, hooks, &geom);
    return (int)R_ref;   // what the reference returns: the tile count of ITS rects (== R with the tight rects off)
                         // == the capacity the binning blob was carved with
}

// Reserved forward: caller-provided blobs, no host wait.  Returns the pass ticket.
long long forward_reserved(char* geom_buffer, char* binning_buffer, char* image_buffer, int reserve, const FwdCall& c)
{
    if (c.P <= 0) return 0;
    if (reserve < 1) throw Error("the pair reservation must be >= 1");
    if (!geom_buffer || !binning_buffer || !image_buffer) throw Error("state buffers must not be NULL");
    validate_forward(c);
    hipStream_t s = c.stream;
    const int dev = current_device();
    FwdPlan plan = make_fwd_plan(c, (uint32_t)reserve);
    if (!plan.generic_depth_sort && take_prefer_generic(dev, c)) plan.generic_depth_sort = 1;
    prepare_depth_bucket_sort(plan.nb);
    const size_t depth_temp = cached_depth_temp((size_t)c.P);
    GeomState geom = GeomState::carve(geom_buffer, (size_t)c.P, depth_temp);
    ImageState img = ImageState::carve(image_buffer, (size_t)c.width * c.height, (size_t)plan.gx * plan.gy);
    BinState bin = BinState::carve(binning_buffer, (size_t)plan.reserve, plan.layout.wide, TileGrid{(uint32_t)plan.gx, (uint32_t)plan.gy});
    PassInfo* info_dev = nullptr;
    const uint64_t ticket = new_ticket(dev, c, plan.reserve, !plan.generic_depth_sort, &info_dev);
    try {
        FwdPassArgs args;
        // the generic (rocPRIM) depth sort has no histogram kernel to carry the header reduction
        fill_fwd_args(args, plan, c, geom, &bin, img, info_dev, ticket, !plan.generic_depth_sort);
        const bool direct = !graphs_enabled() || c.debug || plan.generic_depth_sort || (g_prof.mask.load() & kFwdStages);
        if (direct) {
            FwdPassArgs* d_args;
            {
                std::lock_guard<std::mutex> lk(g_ctx_mu);
                d_args = static_cast<FwdPassArgs*>(direct_block(dev, 0, s, sizeof(FwdPassArgs)));
            }
            StageRun hooks{c.debug != 0};
            issue_forward(plan, d_args, args, s, 3, hooks, &geom);
        } else {
            const CtxKey key{dev, 0, s, c.P, c.M, c.width, c.height, plan.reserve, fwd_flags(plan, c)};
            launch_graph(key, plan, args, s, [](const FwdPlan& p, FwdPassArgs* d, const FwdPassArgs& v, hipStream_t cs) {
                NoHooks nh;
                issue_forward(p, d, v, cs, 3, nh, nullptr);
            });
        }
    } catch (...) {
        forget_ticket(ticket);
        throw;
    }
    return (long long)ticket;
}

template <class F>
int guarded(F&& f)
{
    try {
        g_last_error.clear();
        return f();
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

template <class F>
long long guarded_ll(F&& f)
{
    try {
        g_last_error.clear();
        return f();
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

#define R3_FWD_CALL(coeffs, perband, cumsum, Mval)                                                                      \
    FwdCall c;                                                                                                          \
    c.P = P;                                                                                                            \
    c.D = D;                                                                                                            \
    c.M = (Mval);                                                                                                       \
    c.coeffsNum = (coeffs);                                                                                             \
    c.perBand = (perband);                                                                                              \
    c.cumSum = (cumsum);                                                                                                \
    c.background = background;                                                                                          \
    c.width = width;                                                                                                    \
    c.height = height;                                                                                                  \
    c.means3D = means3D;                                                                                                \
    c.shs = shs;                                                                                                        \
    c.colors_precomp = colors_precomp;                                                                                  \
    c.opacities = opacities;                                                                                            \
    c.scales = scales;                                                                                                  \
    c.scale_modifier = scale_modifier;                                                                                  \
    c.rotations = rotations;                                                                                            \
    c.cov3D_precomp = cov3D_precomp;                                                                                    \
    c.viewmatrix = viewmatrix;                                                                                          \
    c.projmatrix = projmatrix;                                                                                          \
    c.cam_pos = cam_pos;                                                                                                \
    c.tan_fovx = tan_fovx;                                                                                              \
    c.tan_fovy = tan_fovy;                                                                                              \
    c.out_color = out_color;                                                                                            \
    c.out_touched_pixels = out_touched_pixels;                                                                          \
    c.out_transmittance = out_transmittance;                                                                            \
    c.radii = radii;                                                                                                    \
    c.calculate_mean_transmittance = calculate_mean_transmittance;                                                      \
    c.debug = debug;                                                                                                    \
    c.stream = static_cast<hipStream_t>(stream)

void check_ragged(const float* colors_precomp, int bandsNum, const int*& coeffsNum, const int*& perBand, const int*& cumSum)
{
    if (!colors_precomp) {
        if (bandsNum != 4) throw Error("ragged SH path expects 4 bands (degrees 0..3)");
        if (!coeffsNum || !perBand || !cumSum)
            throw Error("ragged SH path needs coeffsNum / perBandPrimitiveCount / cumSumPrimitiveCount");
    } else {
        coeffsNum = perBand = cumSum = nullptr;
    }
}

}  // namespace

int r3::guarded_call(const std::function<int()>& f) { return guarded(f); }

extern "C" {

const char* r3dgs_version(void) { return "r3dgs-hip gfx950 0.4"; }
const char* r3dgs_last_error(void) { return g_last_error.c_str(); }

// The temp-storage part of the geometry blob comes from a rocPRIM query, which needs a visible GPU;
// without one it returns 0 and sets r3dgs_last_error().
size_t r3dgs_geometry_bytes(int P)
{
    try {
        g_last_error.clear();
        return r3::required_bytes<r3::GeomState>((size_t)P, cached_depth_temp((size_t)(P > 0 ? P : 1)));
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return 0;
    }
}
size_t r3dgs_geometry_bytes_lean(int P)
{
    try {
        g_last_error.clear();
        return geometry_blob_bytes((size_t)P, cached_depth_temp((size_t)(P > 0 ? P : 1)), true);
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return 0;
    }
}
size_t r3dgs_binning_bytes(int P, int width, int height, int reserve)
{
    try {
        g_last_error.clear();
        const int gx = (width + r3::kTile - 1) / r3::kTile, gy = (height + r3::kTile - 1) / r3::kTile;
        const r3::PairLayout l = r3::pair_layout(P, (size_t)gx * gy);
        return r3::required_bytes<r3::BinState>((size_t)(reserve > 0 ? reserve : 1), l.wide, r3::TileGrid{(uint32_t)gx, (uint32_t)gy});
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return 0;
    }
}
size_t r3dgs_image_bytes(int width, int height)
{
    const int gx = (width + r3::kTile - 1) / r3::kTile, gy = (height + r3::kTile - 1) / r3::kTile;
    return r3::required_bytes<r3::ImageState>((size_t)width * height, (size_t)gx * gy);
}
int r3dgs_binning_capacity(int P, int width, int height, size_t bytes)
{
    return guarded([&]() {
        const int gx = (width + r3::kTile - 1) / r3::kTile, gy = (height + r3::kTile - 1) / r3::kTile;
        const r3::PairLayout l = r3::pair_layout(P, (size_t)gx * gy);
        if (r3::required_bytes<r3::BinState>((size_t)1, l.wide, r3::TileGrid{(uint32_t)gx, (uint32_t)gy}) > bytes) return 0;
        // largest R whose layout fits: every array size is monotone in R, so any R with the same total has the same
        // offsets -- the capacity recovered from a blob's size reproduces the carve the forward used
        long long lo = 1, hi = 0x7fffffffLL;
        while (lo < hi) {
            const long long mid = (lo + hi + 1) / 2;
            if (r3::required_bytes<r3::BinState>((size_t)mid, l.wide, r3::TileGrid{(uint32_t)gx, (uint32_t)gy}) <= bytes)
                lo = mid;
            else
                hi = mid - 1;
        }
        return (int)lo;
    });
}

int r3dgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                       unsigned char* present, void* stream)
{
    (void)projmatrix;  // unused by the reference too (rasterizer_impl.cu:62-74)
    return guarded([&]() {
        if (P <= 0) return 0;
        if (!means3D || !viewmatrix || !present) throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        r3::launch_mark_visible(P, means3D, viewmatrix, reinterpret_cast<bool*>(present), s);
        r3::check_launch("mark_visible", s, false);
        return 0;
    });
}

int r3dgs_forward(r3dgs_alloc_fn geometryBuffer, void* geometry_user, r3dgs_alloc_fn binningBuffer, void* binning_user,
                  r3dgs_alloc_fn imageBuffer, void* image_user, int P, const int* D, int M, const float* background,
                  int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                  const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                  float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* out_touched_pixels,
                  float* out_transmittance, int* radii, int calculate_mean_transmittance, int debug, void* stream)
{
    (void)prefiltered;
    return guarded([&]() {
        R3_FWD_CALL(nullptr, nullptr, nullptr, M);
        return forward_exact(geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user, c);
    });
}

int r3dgs_inference_forward(r3dgs_alloc_fn geometryBuffer, void* geometry_user, r3dgs_alloc_fn binningBuffer,
                            void* binning_user, r3dgs_alloc_fn imageBuffer, void* image_user, int P, const int* D,
                            int bandsNum, const int* coeffsNum, const int* perBandPrimitiveCount,
                            const int* cumSumPrimitiveCount, const float* background, int width, int height,
                            const float* means3D, const float* shs, const float* colors_precomp,
                            const float* opacities, const float* scales, float scale_modifier,
                            const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                            const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                            int prefiltered, float* out_color, int* out_touched_pixels, float* out_transmittance,
                            int* radii, int calculate_mean_transmittance, int debug, void* stream)
{
    (void)prefiltered;
    return guarded([&]() {
        check_ragged(colors_precomp, bandsNum, coeffsNum, perBandPrimitiveCount, cumSumPrimitiveCount);
        R3_FWD_CALL(coeffsNum, perBandPrimitiveCount, cumSumPrimitiveCount, 16);
        return forward_exact(geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user, c);
    });
}

int r3dgs_reserve_hint(int P, int width, int height)
{
    return guarded([&]() { return (int)reserve_hint(current_device(), P, width, height, 0); });
}

int r3dgs_reserve_hint_view(int P, int width, int height, const float* viewmatrix)
{
    return guarded([&]() { return (int)reserve_hint(current_device(), P, width, height, reinterpret_cast<uintptr_t>(viewmatrix)); });
}

void r3dgs_reserve_forget_view(const float* viewmatrix)
{
    const uintptr_t cam = reinterpret_cast<uintptr_t>(viewmatrix);
    std::lock_guard<std::mutex> lk(g_adv.mu);
    for (auto& v : g_adv.views) v.second.cams.erase(cam);
    // passes of that camera still in flight must not re-enter it when they are harvested
    for (auto& p : g_adv.pending)
        if (p.cam == cam) p.cam = 0;
}

long long r3dgs_forward_reserved(char* geom_buffer, char* binning_buffer, char* image_buffer, int reserve, int P,
                                 const int* D, int M, const float* background, int width, int height,
                                 const float* means3D, const float* shs, const float* colors_precomp,
                                 const float* opacities, const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                 const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                 int prefiltered, float* out_color, int* out_touched_pixels, float* out_transmittance,
                                 int* radii, int calculate_mean_transmittance, int debug, void* stream)
{
    (void)prefiltered;
    return guarded_ll([&]() {
        R3_FWD_CALL(nullptr, nullptr, nullptr, M);
        return forward_reserved(geom_buffer, binning_buffer, image_buffer, reserve, c);
    });
}

long long r3dgs_inference_forward_reserved(char* geom_buffer, char* binning_buffer, char* image_buffer, int reserve,
                                           int P, const int* D, int bandsNum, const int* coeffsNum,
                                           const int* perBandPrimitiveCount, const int* cumSumPrimitiveCount,
                                           const float* background, int width, int height, const float* means3D,
                                           const float* shs, const float* colors_precomp, const float* opacities,
                                           const float* scales, float scale_modifier, const float* rotations,
                                           const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                           const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                                           float* out_color, int* out_touched_pixels, float* out_transmittance,
                                           int* radii, int calculate_mean_transmittance, int debug, void* stream)
{
    (void)prefiltered;
    return guarded_ll([&]() {
        check_ragged(colors_precomp, bandsNum, coeffsNum, perBandPrimitiveCount, cumSumPrimitiveCount);
        R3_FWD_CALL(coeffsNum, perBandPrimitiveCount, cumSumPrimitiveCount, 16);
        return forward_reserved(geom_buffer, binning_buffer, image_buffer, reserve, c);
    });
}

int r3dgs_pass_query(long long ticket, int wait, int* num_rendered, int* visible, int* reserve, int* flags)
{
    return guarded([&]() {
        if (ticket <= 0) throw r3::Error("not a pass ticket");
        const uint64_t t = (uint64_t)ticket, n = ticket_number(t);
        if (g_next_ticket.load() - n >= kInfoRing) throw r3::Error("pass ticket expired (ring reused)");
        // the ring of the device the pass ran on (carried by the ticket), whatever device is current here
        const volatile r3::PassInfo* info = info_ring(ticket_device(t)).host + n % kInfoRing;
        if (wait)
            info = wait_info(t);
        else if (info->seq != (uint32_t)n)
            return 0;
        std::atomic_thread_fence(std::memory_order_acquire);
        const uint32_t R = info->num_rendered, pairs = info->pairs, cap = info->reserve;
        if (num_rendered) *num_rendered = (int)(R > 0x7fffffffu ? 0x7fffffffu : R);
        if (visible) *visible = (int)info->visible;
        if (reserve) *reserve = cap == 0xFFFFFFFFu ? -1 : (int)cap;
        // the depth-bucket hint is stamped by a later kernel of the pass than the header: reported once it is there
        if (flags) *flags = (cap != 0xFFFFFFFFu && pairs > cap ? R3DGS_PASS_TRUNCATED : 0) |
                            (info->sort_seq == (uint32_t)n && info->sort_overflow ? R3DGS_PASS_DEPTH_BUCKET_OVERFLOW : 0);
        return 1;
    });
}

int r3dgs_pass_pairs(long long ticket, int wait)
{
    return guarded([&]() {
        if (ticket <= 0) throw r3::Error("not a pass ticket");
        const uint64_t t = (uint64_t)ticket, n = ticket_number(t);
        if (g_next_ticket.load() - n >= kInfoRing) throw r3::Error("pass ticket expired (ring reused)");
        const volatile r3::PassInfo* info = info_ring(ticket_device(t)).host + n % kInfoRing;
        if (wait)
            info = wait_info(t);
        else if (info->seq != (uint32_t)n)
            throw r3::Error("pass not published yet");
        std::atomic_thread_fence(std::memory_order_acquire);
        const uint32_t pairs = info->pairs;
        return (int)(pairs > 0x7fffffffu ? 0x7fffffffu : pairs);
    });
}

int r3dgs_forward_pairs(void) { return g_last_forward_pairs; }

int r3dgs_set_tight_rects(int on)   // on < 0: query only
{
    const int before = tight_rects();
    if (on >= 0) g_tight_rects.store(on ? 1 : 0);
    return before;
}

void r3dgs_forward_hint(int will_backward) { g_next_forward_trains = will_backward ? 1 : 0; }

int r3dgs_set_sh_cache(int on)   // on < 0: query only
{
    const int before = sh_derivative_cache();
    if (on >= 0) g_sh_cache.store(on ? 1 : 0);
    return before;
}

int r3dgs_set_f64_chain(int on)   // on < 0: query only
{
    const int before = f64_chain();
    if (on >= 0) g_f64_chain.store(on ? 1 : 0);
    return before;
}

int r3dgs_set_bwd_segments(int on)   // on < 0: query only
{
    const int before = bwd_segments();
    if (on >= 0) g_bwd_segments.store(on ? 1 : 0);
    return before;
}

int r3dgs_set_tile_order(int on)   // on < 0: query only
{
    const int before = heaviest_tiles_first();
    if (on >= 0) g_tile_order.store(on ? 1 : 0);
    return before;
}

int r3dgs_export_rects(int P, char* geom_buffer, unsigned short* rects, void* stream)
{
    return guarded([&]() {
        using namespace r3;
        if (P <= 0) return 0;
        if (!geom_buffer || !rects) throw Error("a required pointer is NULL");
        GeomState geom = GeomState::carve(geom_buffer, (size_t)P, cached_depth_temp((size_t)P));
        R3_HIP(hipMemcpyAsync(rects, geom.rect, sizeof(ushort4) * (size_t)P, hipMemcpyDeviceToDevice,
                              static_cast<hipStream_t>(stream)));
        return 0;
    });
}

void r3dgs_reserve_forget(void)
{
    std::lock_guard<std::mutex> lk(g_adv.mu);
    g_adv.views.clear();
    g_adv.pending.clear();
}

long long r3dgs_reserve_overflow_events(int* last_num_rendered, int* last_reserve)
{
    std::lock_guard<std::mutex> lk(g_adv.mu);
    try {
        harvest_locked();
    } catch (...) {
    }
    if (last_num_rendered) *last_num_rendered = (int)g_adv.last_overflow_rendered;
    if (last_reserve) *last_reserve = (int)g_adv.last_overflow_reserve;
    return (long long)g_adv.overflow_events;
}

int r3dgs_backward(int P, const int* D, int M, int R, const float* background, int width, int height,
                   const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                   float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                   const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                   char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix, float* dL_dmean2D,
                   float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                   float* dL_dsh, float* dL_dscale, float* dL_drot, float lambda_sh_sparsity, int debug, void* stream)
{
    return guarded([&]() {
        using namespace r3;
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (P <= 0) return 0;
        if (!geom_buffer || !image_buffer) throw Error("state buffers must not be NULL");
        if (!means3D || !viewmatrix || !projmatrix || !campos || !background || !dL_dpix)
            throw Error("a required pointer is NULL");
        if (!dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D || !dL_dscale || !dL_drot)
            throw Error("a gradient output pointer is NULL");
        if (shs && (!dL_dsh || !D || M < 1 || M > 16)) throw Error("SH gradients need dL_dsh, degrees and 1 <= M <= 16");
        BwdPlan plan;
        plan.P = P;
        plan.M = M;
        plan.W = width;
        plan.H = height;
        plan.gx = (width + kTile - 1) / kTile;
        plan.gy = (height + kTile - 1) / kTile;
        plan.layout = pair_layout(P, (size_t)plan.gx * plan.gy);
        plan.bwd_ppl = 4;   // one wave per tile: the per-pair gradient slab has ONE owner per (tile, Gaussian)
        // R: the pair capacity the forward carved the binning blob with (num_rendered of an exact-size forward, the
        // reservation of a reserved one); the pair count itself is read from the device header
        plan.reserve = R > 0 ? (uint32_t)R : 1u;
        plan.grid_pairs = grid_pairs_for(plan.reserve);
        plan.has_pairs = binning_buffer != nullptr ? 1 : 0;
        plan.units_cap = bwd_units_cap(plan.reserve, TileGrid{(uint32_t)plan.gx, (uint32_t)plan.gy});
        plan.f64_chain = f64_chain();
        const int dev = current_device();
        GeomState geom = GeomState::carve(geom_buffer, (size_t)P, cached_depth_temp((size_t)P));
        ImageState img = ImageState::carve(image_buffer, (size_t)width * height, (size_t)plan.gx * plan.gy);
        BinState bin = BinState::carve(binning_buffer, (size_t)plan.reserve, plan.layout.wide, TileGrid{(uint32_t)plan.gx, (uint32_t)plan.gy});
        if (!radii) radii = geom.radii_internal;

        BwdPassArgs a;
        memset(&a, 0, sizeof(a));
        BlendBwdArgs& bb = a.blend;
        bb.ranges = img.ranges;
        bb.point_list = bin.point_list;
        bb.rec = geom.rec;
        bb.final_T = img.final_T;
        bb.n_contrib = img.n_contrib;
        bb.dL_dpix = dL_dpix;
        bb.W = width;
        bb.H = height;
        bb.gx = plan.gx;
        bb.nblocks = (uint32_t)(plan.gx * plan.gy * (4 / plan.bwd_ppl));
        bb.bg = background;
        bb.pair_grad = bin.pair_grad;
        bb.pair_flag = bin.pair_flag;
        bb.quad_masks = binning_buffer && keep_quad_masks() ? bin.quad_masks : nullptr;
        // the unit order lives in the binning blob: a pass without one (no pair was reserved) has nothing to order; a unit
        // word holds kUnitTileBits of tile index (an image of more than 2^20 tiles -- 268 Mpix -- is walked row-major)
        bb.tile_order = (heaviest_tiles_first() && plan.has_pairs && (size_t)plan.gx * plan.gy <= ((size_t)1 << kUnitTileBits))
                            ? bin.unit_order : nullptr;
        bb.units_cap = plan.units_cap;
        bb.quad_depth = img.quad_depth;
        bb.ckpt = plan.has_pairs ? bin.ckpt : nullptr;
        bb.hdr = geom.header;
        bb.segments = bwd_segments();
        PairReduceArgs& pr = a.reduce;
        pr.hdr = geom.header;
        pr.pair_grad = bin.pair_grad;
        pr.pair_flag = bin.pair_flag;
        pr.pair_rank = bin.pair_rank;
        pr.rank_mask = plan.layout.wide ? 0xFFFFFFFFu : (1u << plan.layout.rank_bits) - 1u;
        pr.order = nullptr;   // the run key is the Gaussian id itself
        pr.rec = geom.rec;
        pr.tiles = geom.tiles;
        pr.acc = geom.acc;
        pr.wave_part = bin.wave_part;
        {   // this pass's stamp: a 64-bit counter that starts at a random value per process -- no row left by an earlier
            // pass of this or of another process (recycled device memory) can carry it
            static std::atomic<unsigned long long> stamp{[] {
                std::random_device rd;
                return ((unsigned long long)rd() << 32) ^ (unsigned long long)rd() ^ 0x9E3779B97F4A7C15ull;
            }()};
            const unsigned long long st = stamp.fetch_add(1);
            pr.stamp0 = (uint32_t)st;
            pr.stamp1 = (uint32_t)(st >> 32);
        }
        PreBwdArgs& pb = a.pre;
        FwdInputs& in = pb.in;
        in.P = P;
        in.M = M;
        in.degrees = D;
        in.means3D = means3D;
        in.scales = scales;
        in.rotations = rotations;
        in.opacities = nullptr;
        in.shs = colors_precomp ? nullptr : shs;
        in.cov3D_precomp = cov3D_precomp;
        in.colors_precomp = colors_precomp;
        ViewParams& view = pb.view;
        view.view = viewmatrix;
        view.proj = projmatrix;
        view.campos = campos;
        view.bg = background;
        view.tan_fovx = tan_fovx;
        view.tan_fovy = tan_fovy;
        view.W = width;
        view.H = height;
        view.scale_modifier = scale_modifier;
        pb.radii = radii;
        pb.rec = geom.rec;
        pb.tiles = geom.tiles;
        pb.acc = geom.acc;
        pb.wave_part = plan.has_pairs ? bin.wave_part : nullptr;
        pb.pair_grad = plan.has_pairs ? bin.pair_grad : nullptr;
        pb.stamp0 = pr.stamp0;
        pb.stamp1 = pr.stamp1;
        pb.header = geom.header;
        pb.lambda_sh = lambda_sh_sparsity;
        pb.sh_ddir = (lambda_sh_sparsity == 0.f && sh_derivative_cache()) ? geom.sh_ddir : nullptr;
        pb.out.dL_dmean2D = dL_dmean2D;
        pb.out.dL_dopacity = dL_dopacity;
        pb.out.dL_dcolor = dL_dcolor;
        pb.out.dL_dmean3D = dL_dmean3D;
        pb.out.dL_dcov3D = dL_dcov3D;
        pb.out.dL_dsh = dL_dsh;
        pb.out.dL_dscale = dL_dscale;
        pb.out.dL_drot = dL_drot;
        pb.out.dL_dconic = dL_dconic;
        {
            static const int stagger = env_int("R3DGS_PREBWD_STAGGER", 127, 0, 4096);   // 0 / 64 / 127 / 254 / 508:
                                                                                         // 84.0 / 79.5 / 79.8 / 79.7 / 88.6 us
            pb.stagger = stagger;
        }

        const bool direct = !graphs_enabled() || debug || (g_prof.mask.load() & kBwdStages);
        if (direct) {
            BwdPassArgs* d_args;
            {
                std::lock_guard<std::mutex> lk(g_ctx_mu);
                d_args = static_cast<BwdPassArgs*>(direct_block(dev, 1, s, sizeof(BwdPassArgs)));
            }
            StageRun hooks{debug != 0};
            issue_backward(plan, d_args, a, s, hooks);
        } else {
            const uint32_t flags = (uint32_t)plan.bwd_ppl | ((uint32_t)plan.has_pairs << 3) | ((uint32_t)plan.layout.wide << 4) |
                                   ((uint32_t)(a.blend.tile_order != nullptr) << 6) |   // one more kernel in the chain
                                   ((uint32_t)plan.f64_chain << 7);
            const CtxKey key{dev, 1, s, P, M, width, height, plan.reserve, flags};
            launch_graph(key, plan, a, s, [](const BwdPlan& p, BwdPassArgs* d, const BwdPassArgs& v, hipStream_t cs) {
                NoHooks nh;
                issue_backward(p, d, v, cs, nh);
            });
        }
        return 0;
    });
}

int r3dgs_colour_variance_accumulate(int P, const int* D, int M, int max_sh_deg, const float* means3D,
                                     const float* cam_pos, const float* shs, const int* radii,
                                     const int* touched_pixels, const float* transmittance, float* wSum, float* wSumSq,
                                     float* mean, float* variance, float* colourDistancesAccum, void* stream)
{
    return guarded([&]() {
        if (P <= 0) return 0;
        if (max_sh_deg < 1 || max_sh_deg > 3) throw r3::Error("max_sh_deg must be in [1,3]");
        if (M < (max_sh_deg + 1) * (max_sh_deg + 1) || M > 16) throw r3::Error("SH tensor too small for max_sh_deg");
        if (!D || !means3D || !cam_pos || !shs || !radii || !touched_pixels || !transmittance || !wSum || !wSumSq ||
            !mean || !variance || !colourDistancesAccum)
            throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        r3::launch_colour_variance_accumulate(P, D, M, max_sh_deg, means3D, cam_pos, shs, radii, touched_pixels,
                                              transmittance, wSum, wSumSq, mean, variance, colourDistancesAccum, s);
        r3::check_launch("colour variance accumulate", s, false);
        return 0;
    });
}

int r3dgs_profile_enable(int on)
{
    // on == 0: off; on == 1: every stage; otherwise bit (s + 1) of `on` selects stage s alone.  A pass with a timed
    // stage is issued with direct launches (events sit between its kernels); the other pass keeps its graph.
    g_prof.mask.store(on == 0 ? 0u : (on == 1 ? ((1u << kNumStages) - 1u) : (((unsigned)on >> 1) & ((1u << kNumStages) - 1u))));
    return 0;
}

int r3dgs_profile_stage_count(void) { return kNumStages; }

const char* r3dgs_profile_stage_name(int stage)
{
    static const char* names[kNumStages] = {"preprocess_fwd", "depth_sort_scan", "tile_binning", "blend_fwd",
                                            "blend_bwd",      "preprocess_bwd",  "sh_color",        "blend_bwd_kernel"};
    return (stage >= 0 && stage < kNumStages) ? names[stage] : "";
}

int r3dgs_profile_read(double* total_ms, int* launches)
{
    return guarded([&]() {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        for (int st = 0; st < kNumStages; st++) {
            double sum = 0.0;
            for (auto& ev : g_prof.used[st]) {
                R3_HIP(hipEventSynchronize(ev.second));
                float ms = 0.f;
                R3_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
                sum += ms;
                g_prof.pool.push_back(ev.first);
                g_prof.pool.push_back(ev.second);
            }
            if (total_ms) total_ms[st] = sum;
            if (launches) launches[st] = (int)g_prof.used[st].size();
            g_prof.used[st].clear();
        }
        return 0;
    });
}

int r3dgs_export_binning(int P, int R, int count, int width, int height, char* geom_buffer, char* binning_buffer,
                         char* image_buffer, uint64_t* keys, uint32_t* point_list, uint32_t* ranges, uint32_t* n_contrib,
                         float* final_T, uint32_t* tiles_touched, void* stream)
{
    return guarded([&]() {
        using namespace r3;
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (P <= 0) return 0;
        const int gx = (width + kTile - 1) / kTile, gy = (height + kTile - 1) / kTile;
        const size_t N = (size_t)width * height, Tn = (size_t)gx * gy;
        GeomState geom = GeomState::carve(geom_buffer, (size_t)P, cached_depth_temp((size_t)P));
        ImageState img = ImageState::carve(image_buffer, N, Tn);
        if (count > R) throw Error("count exceeds the blob's pair capacity");
        if (count > 0 && binning_buffer) {
            const PairLayout l = pair_layout(P, Tn);
            BinState bin = BinState::carve(binning_buffer, (size_t)R, l.wide, TileGrid{(uint32_t)gx, (uint32_t)gy});
            if (keys) launch_export_keys(P, count, R, Tn, bin, geom, keys, s);
            if (point_list) {
                // entries beyond the pairs the pass binned (a caller holding only the reference's num_rendered may ask for
                // them: the exact-size blob is carved for that count) are not list entries: ~0, as the keys
                R3_HIP(hipMemsetAsync(point_list, 0xFF, sizeof(uint32_t) * (size_t)count, s));
                launch_export_point_list(count, bin.point_list, geom.header, point_list, s);
            }
        }
        if (ranges) R3_HIP(hipMemcpyAsync(ranges, img.ranges, sizeof(uint2) * Tn, hipMemcpyDeviceToDevice, s));
        if (n_contrib) R3_HIP(hipMemcpyAsync(n_contrib, img.n_contrib, sizeof(uint32_t) * N, hipMemcpyDeviceToDevice, s));
        if (final_T) R3_HIP(hipMemcpyAsync(final_T, img.final_T, sizeof(float) * N, hipMemcpyDeviceToDevice, s));
        if (tiles_touched)
            R3_HIP(hipMemcpyAsync(tiles_touched, geom.tiles, sizeof(uint32_t) * (size_t)P, hipMemcpyDeviceToDevice, s));
        check_launch("export", s, false);
        return 0;
    });
}

int r3dgs_export_tile_order(int P, int R, int width, int height, char* binning_buffer, char* image_buffer,
                            uint32_t* quad_depth, uint32_t* unit_order, void* stream)
{
    return guarded([&]() {
        using namespace r3;
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (!image_buffer) throw Error("a required pointer is NULL");
        const int gx = (width + kTile - 1) / kTile, gy = (height + kTile - 1) / kTile;
        const size_t Tn = (size_t)gx * gy;
        ImageState img = ImageState::carve(image_buffer, (size_t)width * height, Tn);
        if (quad_depth) R3_HIP(hipMemcpyAsync(quad_depth, img.quad_depth, sizeof(uint32_t) * 4 * Tn, hipMemcpyDeviceToDevice, s));
        if (unit_order) {
            if (!binning_buffer || R <= 0) throw Error("the unit order lives in the binning buffer");
            BinState bin = BinState::carve(binning_buffer, (size_t)R, pair_layout(P, Tn).wide, TileGrid{(uint32_t)gx, (uint32_t)gy});
            R3_HIP(hipMemcpyAsync(unit_order, bin.unit_order, sizeof(uint32_t) * ((size_t)bwd_units_cap((uint32_t)R, TileGrid{(uint32_t)gx, (uint32_t)gy}) + 2 * kOrderLists),
                                  hipMemcpyDeviceToDevice, s));
        }
        check_launch("export", s, false);
        return 0;
    });
}

int r3dgs_bwd_units_cap(int R, int width, int height)
{
    const int gx = (width + r3::kTile - 1) / r3::kTile, gy = (height + r3::kTile - 1) / r3::kTile;
    return (int)r3::bwd_units_cap((uint32_t)(R > 0 ? R : 1), r3::TileGrid{(uint32_t)gx, (uint32_t)gy});
}

}  // extern "C"
