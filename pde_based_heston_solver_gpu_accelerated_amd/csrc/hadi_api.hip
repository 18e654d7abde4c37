// hadi_api.hip -- host side of libhadi: handle, HBM buffers, launches.  C ABI in include/hadi.h.
// There is deliberately no CPU compute path in this file: without a GPU hadi_create fails.
#include "../../include/hadi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hadi_kernels.h"
#include "hadi_plan.h"

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> kev;  // per-launch events (profiling)
    int profiling = 0;
    int cu_count = 256;
    std::string name, arch;
    std::string err;
    hadi_timing timing{};
    // grow-only device buffers
    DevBuf U, Y, LAM, U0, UT;
    DevBuf V, R1, C2;                 // Craig-Sneyd: predictor result and carry-over arrays
    DevBuf rs_tab;                    // paired strips: the pairs' coupling column behind the cyclic reduction, per (instance, v-row, half, lane)
    DevBuf Uf, Yf;                    // fp32-state sweep: the two state arrays as float
    DevBuf scoef, b2row, rowc, a2i, pb, rinv, rwork, ipar, par8;
    DevBuf g_s, g_v, g_ds, g_dv;      // grids owned by the library (staged / broadcast)
    DevBuf src_v, src_dv, sel_a, sel_b, v0_i;
    DevBuf natU, natU0, natOut, prices, status;
    // hipGraph cache for the time loop (2 launches per step: launch-bound for small batches)
    struct GraphEntry { std::string key; hipGraph_t graph; hipGraphExec_t exec; unsigned long long stamp; };
    std::vector<GraphEntry> graphs;
    unsigned long long graph_clock = 0;
    int use_graph = 1;
    int graph_max_melems = 8;  // hipGraph replay for batches of up to this many Mi state elements (hadi_set_tuning "graph_max_melems")
    int use_small = 1;  // LDS-resident one-launch path for small grids
    int small_seq = -1;  // ... European / dividend sweeps on the one-wavefront-per-instance kernel: -1 by batch size, 0 never, 1 always
    int use_amp = 1;    // American sweeps without the lambda_bar array when the payoff depends on s only
    int device_vgrid = 1;  // compute_base_prices / compute_jacobian: v-grids rebuilt per instance on the device
    int sub_batch = 1;     // large batches run sub-batch by sub-batch (run_sweep)
    int small_pairs = -1;  // small-grid sequential kernel with two instances per wavefront: -1 by batch size, 0 never, 1 always
    // Two measured alternatives of the column pass, both opt-in (round 4; neither moves the 16-chunk pass by more than +-2 %:
    // profiles/r04_colpass_ab.txt): the blocks of an instance take their full column tiles interleaved (hadi_pb_tiles), and
    // hadi_pass_b2 -- part of the next tile prefetched into LDS -- instead of hadi_pass_b1 for European sweeps of 9 .. 16 chunks
    int tile_il = 0;
    int col_prefetch = 0;
    int cs_strips = 1;       // Craig-Sneyd row passes on the barrier-free strips where the plan chose strips (0: shared ring, as before round 4)
    int streams = 0;       // hadi_set_tuning "streams": 0 automatic (hadi_plan_row_idle), 1 one stream, 2 two streams side by side
    hipStream_t stream2 = nullptr;                 // the second stream of a two-stream sweep
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    int last_nsub = 1;
    HadiTuning tune;    // kernel-selection overrides (hadi_set_tuning)
    hipEvent_t wait_ev = nullptr;  // hadi_wait_stream
    DevBuf lm31;
    std::string last_path;  // which kernels the last sweep ran (hadi_describe_last_sweep)
    DevBuf div_flag, div_amt, div_pct;
    DevBuf pay_mis;  // American: per-instance payoff-shape flags (hadi_payoff_shape_kernel)
    DevBuf order;    // small-grid path: dispatch order of the instances (multi-maturity batches)
    // Sticky device error word: one int in host-pinned, device-visible memory.  Kernels OR a HADI_DEVERR_* code into it
    // (system-scope atomic, only ever on a failure path); finish_timing reads it after the stream synchronisation every
    // entry point ends with -- no copy, no extra launch -- and turns a non-zero word into HADI_ERR_INTERNAL.
    int *err_host = nullptr, *err_dev = nullptr;
    // Pinned staging arena for the small host-side vectors of a call (per-instance parameters, dividend tables, dispatch order,
    // selectors, status words).  A copy out of pageable memory blocks the host and its source has to outlive it -- every such
    // vector used to cost a stream synchronisation in the middle of a call (three per Jacobian: ~0.15 ms of a 1.7 ms call on
    // the calibration grids).  Copies out of this arena are truly asynchronous; it is rewound at the start of every entry point
    // (each ends with a stream synchronisation, so nothing of the previous call is in flight) and grown there when a call asked
    // for more than it holds (the request that did not fit takes the old copy-and-synchronise path once).
    char *pin = nullptr;
    size_t pin_cap = 0, pin_used = 0, pin_want = 0;
    int pin_dirty = 0;  // copies out of the arena may be in flight (cleared by the stream synchronisation that ends a call)
    int debug_fault = 0;  // test hook (hadi_set_tuning "debug_fault"): HADI_DEBUG_* bits handed to the sweep kernels
    // instance-resident launch (hadi_team_kernel): -1 automatic, 0 never, 1 whenever the shape allows it; team_failed is set
    // when a team could not form or a team barrier timed out once on this handle (the automatic choice then stays away)
    int team_launch = -1, team_failed = 0;
    DevBuf team;
};

// Every GPU entry point runs on the handle's device whatever the caller's current device is (a torch rank that
// never called set_device sits on device 0), and leaves the caller's choice as it found it.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int fail(Ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(c, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((c), HADI_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                      \
    } while (0)

int ensure(Ctx *c, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return HADI_OK;
    if (b.p) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, HADI_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return HADI_OK;
}

template <class T>
T *ptr(DevBuf &b) {
    return static_cast<T *>(b.p);
}

// Start of an entry point: nothing of the previous call is in flight (it ended with a stream synchronisation).
void pin_rewind(Ctx *c) {
    if (c->pin_dirty && c->stream) (void)hipStreamSynchronize(c->stream);  // (a call that left through an error path)
    c->pin_dirty = 0;
    if (c->pin_want > c->pin_cap) {
        if (c->pin) (void)hipHostFree(c->pin);
        c->pin = nullptr; c->pin_cap = 0;
        const size_t want = c->pin_want + c->pin_want / 2 + 4096;
        void *q = nullptr;
        if (hipHostMalloc(&q, want, hipHostMallocDefault) == hipSuccess) { c->pin = static_cast<char *>(q); c->pin_cap = want; }
    }
    c->pin_used = 0;
    c->pin_want = 0;
}
// `bytes` of pinned staging memory, or nullptr if the arena cannot hold them (the caller then takes the synchronising path).
void *pin_alloc(Ctx *c, size_t bytes) {
    const size_t need = (bytes + 63) & ~(size_t)63;
    c->pin_want += need;
    if (!c->pin || c->pin_used + need > c->pin_cap) return nullptr;
    void *q = c->pin + c->pin_used;
    c->pin_used += need;
    return q;
}
// Host -> device copy of a small host vector: through the pinned arena (asynchronous, the source may die at once), or, if it
// does not fit, straight from the caller's memory followed by a stream synchronisation.
int stage_to_device(Ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return HADI_OK;
    void *q = pin_alloc(c, bytes);
    if (q) {
        std::memcpy(q, src, bytes);
        c->pin_dirty = 1;
        HIP_TRY(c, hipMemcpyAsync(dst, q, bytes, hipMemcpyHostToDevice, c->stream));
        return HADI_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HADI_OK;
}

int grid1d(size_t n, int block = 256, int cap = 4096) {
    size_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}

// ---- host grid code (grid.cpp:16-61, grid_pod.hpp:25-87) ---------------------------------------
void sorted_insert_drop_largest(double *v, int n, double x0) {
    // push_back(x0); sort; pop_back on an ascending array of n values
    if (!(x0 < v[n - 1])) return;
    int pos = 0;
    while (pos < n && v[pos] <= x0) pos++;
    for (int k = n - 1; k > pos; k--) v[k] = v[k - 1];
    v[pos] = x0;
}
void build_v(int m2, double V_0, double V, double d, double *vec_v, double *delta_v) {
    const double Delta_eta = (1.0 / m2) * std::asinh(V / d);
    for (int i = 0; i <= m2; i++) vec_v[i] = d * std::sinh(i * Delta_eta);
    sorted_insert_drop_largest(vec_v, m2 + 1, V_0);
    for (int i = 0; i < m2; i++) delta_v[i] = vec_v[i + 1] - vec_v[i];
}

// ---- one batched sweep ----------------------------------------------------------------------------
struct SweepDesc {
    int n = 0;               // instances actually solved (6x the caller's for a Jacobian)
    int m1 = 0, m2 = 0, variant = 0, scheme = 0, prec = 0;
    double theta = 0, r_d = 0, r_f = 0;
    std::vector<double> par8;  // [n][8] rho sigma kappa eta dt N . .
    int Nmax = 0;
    bool uniform_steps = true;
    double dt0 = 0;
    const double *d_vec_s = nullptr, *d_vec_v = nullptr, *d_delta_s = nullptr, *d_delta_v = nullptr;  // device [n][..]
    const double *d_natU = nullptr;   // device natural [n_src][m]
    const double *d_natU0 = nullptr;  // device natural [n_src][m] or null
    int n_src = 0;                    // natural arrays hold n_src instances, instance k reads k % n_src
    int num_div = 0;
    const double *div_dates = nullptr, *div_amounts = nullptr, *div_pcts = nullptr;
    // diagnostics (hadi_debug_*): 1 = only the row pass of step debug_step, 2 = only one column solve of the packed input
    int debug = 0, debug_step = 1;
};

template <int B, int G, int NG, int PD>
void launch_pass_a(const HadiPlan &pl, const HadiSweepArgs &a, int n, hipStream_t s, int mode = 0) {
    if (mode == 1)
        hipLaunchKernelGGL((hadi_pass_a<B, G, 4, NG, PD, false, 1>), dim3(pl.grid_a), dim3(64 * pl.W * G * NG), pl.smem_a, s, a, n);
    else if (mode == 2)
        hipLaunchKernelGGL((hadi_pass_a<B, G, 4, NG, PD, false, 2>), dim3(pl.grid_a), dim3(64 * pl.W * G * NG), pl.smem_a, s, a, n);
    else if (a.american)
        hipLaunchKernelGGL((hadi_pass_a<B, G, 4, NG, PD, true>), dim3(pl.grid_a), dim3(64 * pl.W * G * NG), pl.smem_a, s, a, n);
    else
        hipLaunchKernelGGL((hadi_pass_a<B, G, 4, NG, PD, false>), dim3(pl.grid_a), dim3(64 * pl.W * G * NG), pl.smem_a, s, a, n);
}

// fp32-state row pass: same geometry, the LDS ring holds floats (the coefficient arrays and tables stay double)
template <int B, int G, int NG, int PD>
void launch_pass_a_f32(const HadiPlan &pl, const HadiSweepArgs &a, int n, hipStream_t s) {
    const size_t ring_elems = (size_t)NG * ((PD + 1) * pl.W + 4) * pl.L.rowp;
    const size_t smem = pl.smem_a - ring_elems * (sizeof(double) - sizeof(float));
    hipLaunchKernelGGL((hadi_pass_a<B, G, 4, NG, PD, false, 0, float>), dim3(pl.grid_a), dim3(64 * pl.W * G * NG), smem, s, a, n);
}

// American, P representation (no lambda_bar array): shared-ring kernel with the payoff row behind the tables in LDS
template <int B, int G, int NG, int PD>
void launch_pass_a_amp(const HadiPlan &pl, const HadiSweepArgs &a, int n, hipStream_t s) {
    hipLaunchKernelGGL((hadi_pass_a<B, G, 4, NG, PD, 2>), dim3(pl.grid_a), dim3(64 * pl.W * G * NG),
                       pl.smem_a + (size_t)pl.L.rowp * sizeof(double), s, a, n);
}

// Kernels whose dynamic LDS can exceed the 64 KiB default need the limit raised once.
template <class K>
hipError_t raise_lds_limit(K kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int B, int G, int NG, int PD>
hipError_t raise_pass_a() {
    hipError_t e = raise_lds_limit(hadi_pass_a<B, G, 4, NG, PD, false>);
    if (e == hipSuccess) e = raise_lds_limit(hadi_pass_a<B, G, 4, NG, PD, false, 1>);
    if (e == hipSuccess) e = raise_lds_limit(hadi_pass_a<B, G, 4, NG, PD, false, 2>);
    return e != hipSuccess ? e : raise_lds_limit(hadi_pass_a<B, G, 4, NG, PD, true>);
}
hipError_t raise_all_lds_limits() {
    hipError_t e;
    if ((e = raise_pass_a<1, 1, 1, 2>()) != hipSuccess) return e;
    if ((e = raise_pass_a<2, 1, 1, 2>()) != hipSuccess) return e;
    if ((e = raise_pass_a<4, 1, 1, 2>()) != hipSuccess) return e;
    if ((e = raise_pass_a<8, 1, 1, 1>()) != hipSuccess) return e;
    if ((e = raise_pass_a<8, 2, 1, 1>()) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<8, false>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<8, true>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<4, false>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<4, true>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<2, false>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<2, true>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<8, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<4, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<2, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_pairs<0>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_pairs<1>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_pairs<2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_team_kernel<8>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_team_kernel<4>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_seq2_kernel<1>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_seq2_kernel<2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_seq_kernel<1>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_seq_kernel<2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_kernel<1, 4, false>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_kernel<1, 4, true>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_kernel<2, 4, false>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_kernel<2, 4, true>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_kernel<1, 8, false>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_kernel<1, 8, true>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_kernel<2, 8, false>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_small_kernel<2, 8, true>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_b<8, false>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_b<8, true>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_b1<16, false>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_b2<16, double, HADI_B2_NPF(8)>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_b2<16, float, HADI_B2_NPF(4)>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_b<8, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_b1<16, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a<1, 1, 4, 1, 2, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a<2, 1, 4, 1, 2, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a<4, 1, 4, 1, 2, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a<8, 1, 4, 1, 1, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a<8, 2, 4, 1, 1, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<8, false, float>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<8, false, float, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<8, false, double, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<8, 1, double, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a_strip<8, 2, double, 2>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_b<8, false, float>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_b1<16, false, float>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a<1, 1, 4, 1, 2, false, 0, float>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a<2, 1, 4, 1, 2, false, 0, float>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a<4, 1, 4, 1, 2, false, 0, float>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a<8, 1, 4, 1, 1, false, 0, float>)) != hipSuccess) return e;
    if ((e = raise_lds_limit(hadi_pass_a<8, 2, 4, 1, 1, false, 0, float>)) != hipSuccess) return e;
    return raise_lds_limit(hadi_pass_b1<16, true>);
}

// ---- run_sweep, part 1: how the batch is cut ---------------------------------------------------------------------------
struct SubBatch { int off, cnt; HadiPlan pl; int lane; };  // lane: 0 = the handle's stream, 1 = its second stream
struct BatchPlan {
    std::vector<SubBatch> subs;
    bool two_streams = false;
    int fork_before = 0;  // the second stream forks off right before this sub-batch is enqueued
};
// Sub-batches (whole rounds of one instance per CU + the remainder) and the one-or-two-streams decision.  `pl` is the plan of
// the whole batch on entry and the plan the caller sees (layout, table sizes) on exit.
int plan_batches(Ctx *c, const SweepDesc &d, HadiPlan &pl, int state_bytes, bool seq_shape, BatchPlan &bp) {
    std::vector<SubBatch> &subs = bp.subs;
    bool &two_streams = bp.two_streams;
    int &fork_before = bp.fork_before;
    // Large batches on grids where ONE round of the one-block-per-CU kernels (cu_count instances) already moves more than
    // the 256 MB memory-side cache holds: the two passes of a step then re-use each other's data only while the batch is
    // one round deep (measured at 512x256: 512 instances at once ran the column pass 6 % slower per instance than 256;
    // 384 at once: 0.188 + 0.205 ms per step against 0.173 + 0.177 as 256 + 128).  Instances are independent, so the time
    // loop runs sub-batch by sub-batch -- whole rounds of cu_count instances plus the remainder (a remainder below a
    // quarter round rides with the last full round) -- each with the launch geometry of its own size.
    // (the strip kernels scale the A1 action by (1 - theta) / theta and keep the s-convection weights multiplied by
    // theta dt (r_d - r_f): hadi_strip_step)
    const bool no_strips = !(d.theta > 0.0) || d.r_d == d.r_f;
    auto plan_for = [&](int cnt, HadiPlan *q) {
        if (hadi_make_plan(d.m1, d.m2, cnt, 8 * c->cu_count, q, c->tune, state_bytes)) return 1;
        if (no_strips) q->use_strip = 0;
        return 0;
    };
    if (no_strips) pl.use_strip = 0;
    if (d.scheme == HADI_SCHEME_DOUGLAS && !d.debug && c->sub_batch && d.n > c->cu_count &&
        2ll * c->cu_count * pl.L.inst_stride * (long long)state_bytes >= (256ll << 20)) {  // (bytes the sweep streams: 4 per element with the fp32 state)
        const int cu = c->cu_count, full = d.n / cu, rem = d.n - full * cu;
        for (int k = 0; k < full; k++) subs.push_back(SubBatch{k * cu, cu, pl, 0});
        if (rem >= cu / 4) subs.push_back(SubBatch{full * cu, rem, pl, 0});
        else subs.back().cnt += rem;
        for (auto &sbt : subs)
            if (plan_for(sbt.cnt, &sbt.pl)) return fail(c, HADI_ERR_UNSUPPORTED, "plan failed");
        pl = subs[0].pl;  // (what the caller sees: layout and table sizes are the same for every sub-batch)
    } else {
        subs.push_back(SubBatch{0, d.n, pl, 0});
    }
    // Two streams.  Forced (hadi_set_tuning "streams" = 2): the sub-batches alternate between the two streams from the start; a
    // batch that is one sub-batch is cut in two halves for it.  Automatic ("streams" = 0, the default): the LAST sub-batch --
    // the whole batch, or the remainder behind the full rounds -- is cut in two halves that run side by side when its row
    // pass would leave a partial round of CUs idle (hadi_plan_row_idle); the full rounds before it run on one stream.
    // Instances are independent and the two passes of a step stay ordered within their own stream.
    const bool streams_ok = d.scheme == HADI_SCHEME_DOUGLAS && !d.debug && !c->profiling && d.n >= 2 && !seq_shape;
    auto split_last = [&]() -> int {
        const SubBatch last = subs.back();
        const int h0 = (last.cnt + 1) / 2;
        subs.pop_back();
        subs.push_back(SubBatch{last.off, h0, last.pl, 0});
        subs.push_back(SubBatch{last.off + h0, last.cnt - h0, last.pl, 1});
        for (size_t k = subs.size() - 2; k < subs.size(); k++)
            if (plan_for(subs[k].cnt, &subs[k].pl)) return 1;
        return 0;
    };
    if (streams_ok && c->streams == 2) {
        if (subs.size() == 1) {
            if (split_last()) return fail(c, HADI_ERR_UNSUPPORTED, "plan failed");
        } else {
            for (size_t k = 0; k < subs.size(); k++) subs[k].lane = (int)(k & 1);
        }
        two_streams = true;
        fork_before = 0;
    } else if (streams_ok && c->streams == 0 && subs.back().cnt >= 2 &&
               hadi_plan_row_idle(subs.back().pl, subs.back().cnt, c->cu_count) >= HADI_TWO_STREAM_IDLE) {
        const SubBatch whole = subs.back();
        if (split_last()) return fail(c, HADI_ERR_UNSUPPORTED, "plan failed");
        if (subs[subs.size() - 2].pl.use_strip && subs.back().pl.use_strip) {
            two_streams = true;
            fork_before = (int)subs.size() - 2;
        } else {  // (a half that falls back to the shared ring: the rounds argument does not carry over -- one stream)
            subs.pop_back();
            subs.back() = whole;
        }
    }
    // Several sub-batches (whole rounds plus a remainder) and no half-cut above: they alternate between the two streams, as in the
    // forced mode -- the remainder's launches run in the shadow of a full round's instead of behind it.  Measured on strips
    // (profiles/r04_stream_big.txt): 512x256 x320 +5.1 %, x384 +3.2 %, American x320 / x384 +5.8 / +5.9 %, and within +-1 % from
    // two full rounds on (x512 +0.7 %, x768 -0.3 %, x1024 +0.8 %): never a loss, deterministic per (shape, batch size).
    if (streams_ok && c->streams == 0 && !two_streams && subs.size() >= 2) {
        bool strips = true;
        for (auto &sb : subs) strips = strips && sb.pl.use_strip;
        if (strips) {
            for (size_t k = 0; k < subs.size(); k++) subs[k].lane = (int)(k & 1);
            two_streams = true;
            fork_before = 0;
        }
    }
    if (two_streams) pl = subs[0].pl;
    return HADI_OK;
}

// ---- run_sweep, part 2: which kernel runs a pass -----------------------------------------------------------------------
// Everything the choice depends on (the selection rules themselves: hadi_plan.h and DESIGN.md section 4.1).
struct PassEnv {
    const HadiPlan &pl;      // launch geometry of THIS sub-batch
    const HadiLayout &L;
    int nsb;                 // instances of the sub-batch
    hipStream_t q;
    int nstep;
    bool american, amp, xstep, f32;  // amp: P representation; xstep: this step runs on the explicit (U, lambda_bar) pair
    int col_prefetch;
    int cs_strips;                   // Craig-Sneyd row passes on strips where the plan chose strips (tuning key "cs_strips", default on)
};
// Row pass of one time step.  mode: 0 Douglas, 1 / 2 Craig-Sneyd predictor / corrector.
void launch_row_pass(const PassEnv &e, const HadiSweepArgs &ar, int mode) {
    const HadiPlan &pl = e.pl; const HadiLayout &L = e.L; const int nsb = e.nsb, nstep = e.nstep; hipStream_t q = e.q;
    const bool american = e.american, amp = e.amp, xstep = e.xstep, f32 = e.f32;
    if (pl.row_seq) {  // more than 1024 s-intervals: one lane per v-row, sequential along s
        const dim3 g((unsigned)(nsb * ((L.nrows + 63) / 64))), b(64);
        if (american) hipLaunchKernelGGL((hadi_pass_a_seq<1>), g, b, 0, q, ar, nstep);
        else hipLaunchKernelGGL((hadi_pass_a_seq<0>), g, b, 0, q, ar, nstep);
        return;
    }
    if (pl.use_pairs && pl.use_strip && mode == 0 && !f32) {  // 4 nodes per lane: two strips per wavefront
        const dim3 g(pl.grid_as), b(64 * HADI_PAIR_WAVES);
        if (amp && !xstep) hipLaunchKernelGGL((hadi_pass_a_pairs<2>), g, b, pl.smem_pairs_amp, q, ar, nstep);
        else if (american) hipLaunchKernelGGL((hadi_pass_a_pairs<1>), g, b, pl.smem_pairs_eu, q, ar, nstep);
        else hipLaunchKernelGGL((hadi_pass_a_pairs<0>), g, b, pl.smem_pairs_eu, q, ar, nstep);
        return;
    }
    if (amp && !xstep && pl.use_strip && mode == 0) {  // P representation on barrier-free strips
        const dim3 g(pl.grid_as), b(64 * HADI_STRIP_WAVES(L.B));
        const size_t sm = pl.smem_as + (size_t)L.rowp * sizeof(double);  // + the payoff row
        if (L.G == 2) {  // paired strips (two wavefronts per row)
            hipLaunchKernelGGL((hadi_pass_a_strip<8, 2, double, 2>), g, b, sm, q, ar, nstep);
            return;
        }
        switch (L.B) {
            case 8: hipLaunchKernelGGL((hadi_pass_a_strip<8, 2>), g, b, sm, q, ar, nstep); break;
            case 4: hipLaunchKernelGGL((hadi_pass_a_strip<4, 2>), g, b, sm, q, ar, nstep); break;
            default: hipLaunchKernelGGL((hadi_pass_a_strip<2, 2>), g, b, sm, q, ar, nstep); break;
        }
        return;
    }
    if (amp && !xstep) {
        switch (L.B * 10 + L.G) {
            case 11: launch_pass_a_amp<1, 1, 1, 2>(pl, ar, nstep, q); break;
            case 21: launch_pass_a_amp<2, 1, 1, 2>(pl, ar, nstep, q); break;
            case 41: launch_pass_a_amp<4, 1, 1, 2>(pl, ar, nstep, q); break;
            case 81: launch_pass_a_amp<8, 1, 1, 1>(pl, ar, nstep, q); break;
            default: launch_pass_a_amp<8, 2, 1, 1>(pl, ar, nstep, q); break;
        }
        return;
    }
    if (f32 && pl.use_strip && L.B == 8 && L.G == 2) {  // fp32 state, 512 < m1 <= 1024: paired strips
        hipLaunchKernelGGL((hadi_pass_a_strip<8, false, float, 2>), dim3(pl.grid_as), dim3(512), pl.smem_as, q, ar, nstep);
        return;
    }
    if (f32 && pl.use_strip && L.B == 8) {  // fp32 state, 8 nodes per lane, large batch: strips with a ring of floats
        const size_t smem = (size_t)8 * 4 * L.rowp * sizeof(float) + (size_t)4 * 64 * L.B * sizeof(double);
        hipLaunchKernelGGL((hadi_pass_a_strip<8, false, float>), dim3(pl.grid_as), dim3(512), smem, q, ar, nstep);
        return;
    }
    if (f32) {  // fp32 state: shared-ring kernel for the other shapes
        switch (L.B * 10 + L.G) {
            case 11: launch_pass_a_f32<1, 1, 1, 2>(pl, ar, nstep, q); break;
            case 21: launch_pass_a_f32<2, 1, 1, 2>(pl, ar, nstep, q); break;
            case 41: launch_pass_a_f32<4, 1, 1, 2>(pl, ar, nstep, q); break;
            case 81: launch_pass_a_f32<8, 1, 1, 1>(pl, ar, nstep, q); break;
            default: launch_pass_a_f32<8, 2, 1, 1>(pl, ar, nstep, q); break;
        }
        return;
    }
    if (pl.use_strip && mode != 0 && !pl.use_pairs && (e.cs_strips == 1 || e.cs_strips == 1 + mode)) {  // (2 / 3: diagnostics -- only the predictor / only the corrector on strips)  // Craig-Sneyd predictor / corrector on strips (European, fp64)
        const dim3 g(pl.grid_as), b(64 * HADI_STRIP_WAVES(L.B));
        if (L.G == 2) {
            if (mode == 1) hipLaunchKernelGGL((hadi_pass_a_strip<8, 0, double, 2, 1>), g, b, pl.smem_as, q, ar, nstep);
            else hipLaunchKernelGGL((hadi_pass_a_strip<8, 0, double, 2, 2>), g, b, pl.smem_as, q, ar, nstep);
            return;
        }
        switch (L.B * 4 + mode) {
            case 33: hipLaunchKernelGGL((hadi_pass_a_strip<8, 0, double, 1, 1>), g, b, pl.smem_as, q, ar, nstep); break;
            case 34: hipLaunchKernelGGL((hadi_pass_a_strip<8, 0, double, 1, 2>), g, b, pl.smem_as, q, ar, nstep); break;
            case 17: hipLaunchKernelGGL((hadi_pass_a_strip<4, 0, double, 1, 1>), g, b, pl.smem_as, q, ar, nstep); break;
            case 18: hipLaunchKernelGGL((hadi_pass_a_strip<4, 0, double, 1, 2>), g, b, pl.smem_as, q, ar, nstep); break;
            case 9: hipLaunchKernelGGL((hadi_pass_a_strip<2, 0, double, 1, 1>), g, b, pl.smem_as, q, ar, nstep); break;
            default: hipLaunchKernelGGL((hadi_pass_a_strip<2, 0, double, 1, 2>), g, b, pl.smem_as, q, ar, nstep); break;
        }
        return;
    }
    if (pl.use_strip && mode == 0 && L.G == 2) {  // paired strips (Douglas step, two wavefronts per row)
        if (american) hipLaunchKernelGGL((hadi_pass_a_strip<8, 1, double, 2>), dim3(pl.grid_as), dim3(512), pl.smem_as, q, ar, nstep);
        else hipLaunchKernelGGL((hadi_pass_a_strip<8, false, double, 2>), dim3(pl.grid_as), dim3(512), pl.smem_as, q, ar, nstep);
        return;
    }
    if (pl.use_strip && mode == 0) {  // barrier-free strips (Douglas step, one wavefront per row)
        const dim3 g(pl.grid_as), b(64 * HADI_STRIP_WAVES(L.B));
        switch (L.B * 2 + (american ? 1 : 0)) {
            case 16: hipLaunchKernelGGL((hadi_pass_a_strip<8, false>), g, b, pl.smem_as, q, ar, nstep); break;
            case 17: hipLaunchKernelGGL((hadi_pass_a_strip<8, true>), g, b, pl.smem_as, q, ar, nstep); break;
            case 8: hipLaunchKernelGGL((hadi_pass_a_strip<4, false>), g, b, pl.smem_as, q, ar, nstep); break;
            case 9: hipLaunchKernelGGL((hadi_pass_a_strip<4, true>), g, b, pl.smem_as, q, ar, nstep); break;
            case 4: hipLaunchKernelGGL((hadi_pass_a_strip<2, false>), g, b, pl.smem_as, q, ar, nstep); break;
            default: hipLaunchKernelGGL((hadi_pass_a_strip<2, true>), g, b, pl.smem_as, q, ar, nstep); break;
        }
        return;
    }
    switch (L.B * 10 + L.G) {
        case 11: launch_pass_a<1, 1, 1, 2>(pl, ar, nstep, q, mode); break;
        case 21: launch_pass_a<2, 1, 1, 2>(pl, ar, nstep, q, mode); break;
        case 41: launch_pass_a<4, 1, 1, 2>(pl, ar, nstep, q, mode); break;
        case 81: launch_pass_a<8, 1, 1, 1>(pl, ar, nstep, q, mode); break;
        default: launch_pass_a<8, 2, 1, 1>(pl, ar, nstep, q, mode); break;
    }
}

// Column pass of one time step.
void launch_col_pass(const PassEnv &e, const HadiSweepArgs &ar) {
    const HadiPlan &pl = e.pl; const HadiLayout &L = e.L; const int nsb = e.nsb, nstep = e.nstep; hipStream_t q = e.q;
    const bool american = e.american, amp = e.amp, xstep = e.xstep, f32 = e.f32;
    // up to 8 chunks: 512-thread blocks with two register buffers (2 waves per SIMD); 9..16 chunks: the
    // 1024-thread block leaves 128 VGPRs per lane, which only the single-buffer kernel fits
    // (measured at 1024x512: 0.250 vs 0.382 ms/launch for the double-buffered code, which spills)
    const dim3 g(pl.grid_b), b(pl.block_b);
    if (pl.col_seq) {  // more than 16 chunks of v-rows: one lane per storage column, sequential along v
        const dim3 gs((unsigned)(nsb * pl.ctiles)), bs(64);
        if (american) hipLaunchKernelGGL((hadi_pass_b_seq<1>), gs, bs, 0, q, ar, nstep);
        else hipLaunchKernelGGL((hadi_pass_b_seq<0>), gs, bs, 0, q, ar, nstep);
        return;
    }
    if (amp && !xstep) {
        if (L.P <= 8) hipLaunchKernelGGL((hadi_pass_b<8, 2>), g, b, pl.smem_b, q, ar, nstep);
        else hipLaunchKernelGGL((hadi_pass_b1<16, 2>), g, b, pl.smem_b, q, ar, nstep);
        return;
    }
    if (f32) {
        if (L.P <= 8) hipLaunchKernelGGL((hadi_pass_b<8, false, float>), g, b, pl.smem_b, q, ar, nstep);
        else if (e.col_prefetch) hipLaunchKernelGGL((hadi_pass_b2<16, float, HADI_B2_NPF(4)>), g, b, pl.smem_b2, q, ar, nstep);
        else hipLaunchKernelGGL((hadi_pass_b1<16, false, float>), g, b, pl.smem_b, q, ar, nstep);
        return;
    }
    if (L.P <= 8) {
        if (american) hipLaunchKernelGGL((hadi_pass_b<8, true>), g, b, pl.smem_b, q, ar, nstep);
        else hipLaunchKernelGGL((hadi_pass_b<8, false>), g, b, pl.smem_b, q, ar, nstep);
    } else {
        if (american) hipLaunchKernelGGL((hadi_pass_b1<16, true>), g, b, pl.smem_b, q, ar, nstep);
        else if (e.col_prefetch) hipLaunchKernelGGL((hadi_pass_b2<16, double, HADI_B2_NPF(8)>), g, b, pl.smem_b2, q, ar, nstep);
        else hipLaunchKernelGGL((hadi_pass_b1<16, false>), g, b, pl.smem_b, q, ar, nstep);
    }
}

// ---- run_sweep, part 3: the kernels of the streaming path in words (hadi_describe_last_sweep) ------------------------
std::string describe_streaming_path(const Ctx *c, const HadiPlan &pl, const BatchPlan &bp, bool american, bool amp, bool cs, bool f32) {
    const HadiLayout &L = pl.L;
    const std::vector<SubBatch> &subs = bp.subs;
    const int nsub = (int)subs.size();
    const bool two_streams = bp.two_streams;
    const int fork_before = bp.fork_before;
    std::string last_path;
    char buf[256];
    char rowk[96];
    if (amp && pl.use_strip && !cs && L.G == 2) std::snprintf(rowk, sizeof rowk, "hadi_pass_a_strip<8,AM-P,double,2> (paired strips of %d rows, no lambda_bar array)", pl.RS);
    else if (amp && pl.use_strip && !cs) std::snprintf(rowk, sizeof rowk, "hadi_pass_a_strip<%d,AM-P> (strips of %d rows, no lambda_bar array)", L.B, pl.RS);
    else if (amp) std::snprintf(rowk, sizeof rowk, "hadi_pass_a<%d,%d,%d,%d,%d,AM-P> (tiles of %d rows, no lambda_bar array)", L.B, L.G, pl.W, pl.NG, pl.PD, pl.R);
    else if (f32 && pl.use_strip && L.B == 8 && L.G == 2) std::snprintf(rowk, sizeof rowk, "hadi_pass_a_strip<8,EU,float,2> (paired strips of %d rows, fp32 state)", pl.RS);
    else if (f32 && pl.use_strip && L.B == 8) std::snprintf(rowk, sizeof rowk, "hadi_pass_a_strip<8,EU,float> (strips of %d rows, fp32 state)", pl.RS);
    else if (f32) std::snprintf(rowk, sizeof rowk, "hadi_pass_a<%d,%d,%d,%d,%d,EU,float> (tiles of %d rows, fp32 state)", L.B, L.G, pl.W, pl.NG, pl.PD, pl.R);
    else if (pl.use_strip && cs && c->cs_strips && !pl.use_pairs && L.G == 2) std::snprintf(rowk, sizeof rowk, "hadi_pass_a_strip<8,EU,double,2,CS> (paired strips of %d rows)", pl.RS);
    else if (pl.use_strip && cs && c->cs_strips && !pl.use_pairs) std::snprintf(rowk, sizeof rowk, "hadi_pass_a_strip<%d,EU,double,1,CS> (strips of %d rows)", L.B, pl.RS);
    else if (pl.use_strip && !cs && L.G == 2) std::snprintf(rowk, sizeof rowk, "hadi_pass_a_strip<8,%s,double,2> (paired strips of %d rows)", american ? "AM" : "EU", pl.RS);
    else if (pl.use_strip && !cs) std::snprintf(rowk, sizeof rowk, "hadi_pass_a_strip<%d,%s> (strips of %d rows)", L.B, american ? "AM" : "EU", pl.RS);
    else std::snprintf(rowk, sizeof rowk, "hadi_pass_a<%d,%d,%d,%d,%d,%s%s> (tiles of %d rows)", L.B, L.G, pl.W, pl.NG, pl.PD,
                       american ? "AM" : "EU", cs ? ",CS" : "", pl.R);
    if (pl.use_pairs && pl.use_strip && !cs && !f32)
        std::snprintf(rowk, sizeof rowk, "hadi_pass_a_pairs<%s> (two strips of %d rows per wavefront%s)", amp ? "AM-P" : american ? "AM" : "EU", pl.RS,
                      amp ? ", no lambda_bar array" : "");
    if (pl.row_seq) std::snprintf(rowk, sizeof rowk, "hadi_pass_a_seq<%s> (one lane per v-row, sequential along s)", american ? "AM" : "EU");
    if (pl.col_seq)
        std::snprintf(buf, sizeof buf, "row pass %s; column pass hadi_pass_b_seq<%s> (one lane per column, sequential along v)", rowk, american ? "AM" : "EU");
    else
        std::snprintf(buf, sizeof buf, "row pass %s; column pass %s<%d,%s> (%d chunks of %d rows, %d column tiles per block)", rowk,
                      L.P <= 8 ? "hadi_pass_b" : (!american && c->col_prefetch) ? "hadi_pass_b2" : "hadi_pass_b1", L.P <= 8 ? 8 : 16,
                      amp ? "AM-P" : american ? "AM" : "EU", L.P, HADI_LC, pl.btpw);
    last_path = buf;
    if (nsub > 1) {
        bool same = true;
        for (auto &sbt : subs) same = same && sbt.cnt == subs[0].cnt;
        if (same) last_path += "; " + std::to_string(nsub) + " sub-batches of " + std::to_string(subs[0].cnt) + " instances";
        else {
            last_path += "; " + std::to_string(nsub) + " sub-batches of";
            for (auto &sbt : subs) last_path += " " + std::to_string(sbt.cnt);
            last_path += " instances (each with the geometry of its own size)";
        }
        if (two_streams && fork_before > 0) last_path += ", the last two side by side on two streams";
        else if (two_streams) last_path += ", side by side on two streams";
    }
    return last_path;
}

int run_sweep(Ctx *c, const SweepDesc &d, HadiPlan &pl) {
    const int state_bytes = d.prec == HADI_STATE_FP32 ? 4 : 8;
    if (hadi_make_plan(d.m1, d.m2, d.n, 8 * c->cu_count, &pl, c->tune, state_bytes))
        return fail(c, HADI_ERR_UNSUPPORTED, "grid %dx%d not supported (need m1 >= 2, m2 >= 3 and (m1 + 16)(m2 + 1) < 2^28)", d.m1, d.m2);
    const bool seq_shape = pl.row_seq || pl.col_seq;  // shapes beyond the streaming kernels: the sequential passes
    if (seq_shape && (d.scheme != HADI_SCHEME_DOUGLAS || d.prec != HADI_STATE_FP64))
        return fail(c, HADI_ERR_UNSUPPORTED, "grids with m1 > 1024 or m2 > %d run Douglas sweeps with the fp64 state only", HADI_MAX_P * HADI_LC - 1);
    BatchPlan bp;
    {
        const int rcp = plan_batches(c, d, pl, state_bytes, seq_shape, bp);
        if (rcp) return rcp;
    }
    const std::vector<SubBatch> &subs = bp.subs;
    const bool two_streams = bp.two_streams;
    const int fork_before = bp.fork_before;
    const int nsub = (int)subs.size();
    const HadiLayout &L = pl.L;
    const bool american = d.variant == HADI_AM || d.variant == HADI_AM_DIV;
    const bool dividend = d.variant == HADI_DIV || d.variant == HADI_AM_DIV;
    const size_t st = (size_t)L.inst_stride * d.n * sizeof(double);
    int rc;
    if ((rc = ensure(c, c->U, st))) return rc;
    if ((rc = ensure(c, c->Y, st))) return rc;
    if (american) {
        if ((rc = ensure(c, c->LAM, st))) return rc;
        if ((rc = ensure(c, c->U0, st))) return rc;
    }
    if (dividend && (rc = ensure(c, c->UT, st))) return rc;
    const bool cs = d.scheme == HADI_SCHEME_CRAIG_SNEYD;
    const bool f32 = d.prec == HADI_STATE_FP32;  // European Douglas (with or without dividends) only (validated)
    if (f32 && ((rc = ensure(c, c->Uf, st / 2)) || (rc = ensure(c, c->Yf, st / 2)))) return rc;
    if (cs && ((rc = ensure(c, c->V, st)) || (rc = ensure(c, c->R1, st)) || (rc = ensure(c, c->C2, st)))) return rc;
    if (pl.row_seq && (rc = ensure(c, c->R1, st))) return rc;  // (hadi_pass_a_seq parks the Thomas multipliers there)
    const bool pair_tab = L.G == 2 && !cs && !pl.row_seq;  // paired strips (Douglas steps) take the pairs' coupling column from a table built once per solve
    if (pair_tab && (rc = ensure(c, c->rs_tab, (size_t)d.n * L.nrows * 128 * 8))) return rc;
    const size_t n = d.n;
    if ((rc = ensure(c, c->scoef, pl.n_scoef * n * 8))) return rc;
    if ((rc = ensure(c, c->b2row, pl.n_b2row * n * 8))) return rc;
    if ((rc = ensure(c, c->rowc, pl.n_rowc * n * 8))) return rc;
    if ((rc = ensure(c, c->a2i, pl.n_a2i * n * 8))) return rc;
    if ((rc = ensure(c, c->pb, pl.n_pb * n * 8))) return rc;
    if ((rc = ensure(c, c->rinv, pl.n_rinv * n * 8))) return rc;
    if ((rc = ensure(c, c->rwork, pl.n_rwork * n * 8))) return rc;
    if ((rc = ensure(c, c->ipar, sizeof(HadiInstPar) * n))) return rc;
    if ((rc = ensure(c, c->par8, 8 * 8 * n))) return rc;

    hipStream_t s = c->stream;
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    if ((rc = stage_to_device(c, c->par8.p, d.par8.data(), 8 * 8 * n))) return rc;
    // Discrete dividends: host-built table "which dividend does instance k pay at the start of step n" (one shared
    // row when the batch has a single (N, delta_t)), plus the set of steps where anybody pays.
    std::vector<int> div_flags;
    std::vector<char> div_step(d.Nmax + 1, 0);
    const int flag_stride = d.uniform_steps ? 0 : d.Nmax;
    const bool have_div = dividend && d.num_div > 0 && !d.debug;  // (diagnostics take p->U as the state the pass starts from)
    if (have_div) {
        const int rows = d.uniform_steps ? 1 : d.n;
        div_flags.resize((size_t)rows * d.Nmax);
        for (int k = 0; k < rows; k++) {
            const double dt = d.par8[(size_t)k * 8 + 4];
            const int N = (int)d.par8[(size_t)k * 8 + 5];
            int *f = div_flags.data() + (size_t)k * d.Nmax;
            hadi_dividend_steps(N, dt, d.num_div, d.div_dates, f, d.Nmax);
            for (int q = 0; q < d.Nmax; q++)
                if (f[q] >= 0) div_step[q + 1] = 1;
        }
        if ((rc = ensure(c, c->div_flag, div_flags.size() * sizeof(int))) || (rc = ensure(c, c->div_amt, d.num_div * 8)) ||
            (rc = ensure(c, c->div_pct, d.num_div * 8)))
            return rc;
        if ((rc = stage_to_device(c, c->div_flag.p, div_flags.data(), div_flags.size() * sizeof(int))) ||
            (rc = stage_to_device(c, c->div_amt.p, d.div_amounts, (size_t)d.num_div * 8)) ||
            (rc = stage_to_device(c, c->div_pct.p, d.div_pcts, (size_t)d.num_div * 8)))
            return rc;
    }
    // identity padding rows of Y must read as zeros in the column pass (the row pass never writes them)
    HIP_TRY(c, hipMemsetAsync(c->Y.p, 0, st, s));

    HadiSetupArgs sa;
    sa.L = L; sa.n_inst = d.n;
    sa.vec_s = d.d_vec_s; sa.vec_v = d.d_vec_v; sa.delta_s = d.d_delta_s; sa.delta_v = d.d_delta_v;
    sa.par = ptr<double>(c->par8);
    sa.r_d = d.r_d; sa.r_f = d.r_f; sa.theta = d.theta;
    sa.scoef = ptr<double>(c->scoef); sa.b2row = ptr<double>(c->b2row); sa.rowc = ptr<double>(c->rowc);
    sa.a2i = ptr<double>(c->a2i); sa.pb = ptr<double>(c->pb); sa.rinv = ptr<double>(c->rinv);
    sa.rwork = ptr<double>(c->rwork); sa.ipar = ptr<HadiInstPar>(c->ipar);
    hipLaunchKernelGGL(hadi_setup_kernel, dim3(d.n), dim3(256), 0, s, sa);

    const size_t tot = (size_t)L.inst_stride * d.n;
    hipLaunchKernelGGL(hadi_pack_kernel, dim3(grid1d(tot)), dim3(256), 0, s, L, d.n, d.n_src, d.d_natU, ptr<double>(c->U));
    if (american) {
        hipLaunchKernelGGL(hadi_pack_kernel, dim3(grid1d(tot)), dim3(256), 0, s, L, d.n, d.n_src,
                           d.d_natU0 ? d.d_natU0 : d.d_natU, ptr<double>(c->U0));
        HIP_TRY(c, hipMemsetAsync(c->LAM.p, 0, st, s));  // lambda_bar <- 0, device_solver.hpp:310-313
        if ((rc = ensure(c, c->pay_mis, sizeof(int) * n))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->pay_mis.p, 0, sizeof(int) * n, s));
        hipLaunchKernelGGL(hadi_payoff_shape_kernel, dim3(grid1d(tot)), dim3(256), 0, s, L, d.n, ptr<double>(c->U0), ptr<int>(c->pay_mis));
    }
    // American in the P representation (hadi_row_step, AMER == 2): every payoff of the batch must depend on s only.
    // One small device-to-host copy per solve decides it.
    bool amp = false;
    const bool takes_small_path = c->use_small && !c->profiling && !cs && !f32 && !d.debug && (american ? pl.smem_small_am : pl.smem_small_eu) > 0;
    if (american && c->use_amp && !cs && !takes_small_path && !d.debug && !seq_shape) {
        std::vector<int> mis(d.n);
        HIP_TRY(c, hipMemcpyAsync(mis.data(), c->pay_mis.p, sizeof(int) * n, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        amp = true;
        for (int k = 0; k < d.n; k++) amp = amp && mis[k] == 0;
    }
    HIP_TRY(c, hipGetLastError());

    if (f32) {  // round the packed state to fp32; Y's identity padding rows must read as zeros
        hipLaunchKernelGGL(hadi_narrow_kernel, dim3(grid1d(tot)), dim3(256), 0, s, L, ptr<double>(c->U), ptr<float>(c->Uf), tot);
        HIP_TRY(c, hipMemsetAsync(c->Yf.p, 0, st / 2, s));
    }
    HadiSweepArgs a;
    a.U = ptr<double>(c->U); a.Y = ptr<double>(c->Y);
    if (f32) {  // the kernels instantiated for float reinterpret these two
        a.U = reinterpret_cast<double *>(c->Uf.p);
        a.Y = reinterpret_cast<double *>(c->Yf.p);
    }
    a.LAM = american ? ptr<double>(c->LAM) : nullptr;
    a.U0 = american ? ptr<double>(c->U0) : nullptr;
    a.pay_mis = american ? ptr<int>(c->pay_mis) : nullptr;
    a.scoef = ptr<double>(c->scoef); a.b2row = ptr<double>(c->b2row); a.rowc = ptr<double>(c->rowc);
    a.pb = ptr<double>(c->pb); a.rinv = ptr<double>(c->rinv); a.ipar = ptr<HadiInstPar>(c->ipar);
    a.L = L; a.n_inst = d.n; a.R = pl.R; a.ntiles = pl.ntiles; a.ctiles = pl.ctiles; a.btpw = pl.btpw; a.bgroups = pl.bgroups;
    a.american = american ? 1 : 0; a.pos_m1 = pl.pos_m1;
    a.tile_il = c->tile_il;
    a.RS = pl.RS; a.sblocks = pl.sblocks;
    a.err = c->err_dev; a.debug = c->debug_fault;
    a.R1 = (cs || pl.row_seq) ? ptr<double>(c->R1) : nullptr;
    a.C2 = cs ? ptr<double>(c->C2) : nullptr;
    a.rs_tab = pair_tab ? ptr<double>(c->rs_tab) : nullptr;
    // Craig-Sneyd: the predictor's column pass writes V (= Y2), the corrector's row pass reads V
    HadiSweepArgs av = a;
    if (cs) av.U = ptr<double>(c->V);

    const bool prof = c->profiling != 0 && !d.debug;
    if (prof) {
        const size_t need = (size_t)4 * d.Nmax * nsub;
        while (c->kev.size() < need) {
            hipEvent_t e;
            HIP_TRY(c, hipEventCreate(&e));
            c->kev.push_back(e);
        }
    }
    // The whole time loop as a function of the stream, so it can be enqueued directly or captured.
    // Instance offset `o` applied to every per-instance array of the sweep arguments (sub-batches, see nsub above).
    auto shift = [&](HadiSweepArgs x, int o, int cnt, const HadiPlan &sp) {
        x.n_inst = cnt; x.R = sp.R; x.ntiles = sp.ntiles; x.ctiles = sp.ctiles; x.btpw = sp.btpw; x.bgroups = sp.bgroups;
        x.RS = sp.RS; x.sblocks = sp.sblocks;
        const size_t so = (size_t)o * L.inst_stride;
        if (f32) {
            x.U = reinterpret_cast<double *>(reinterpret_cast<float *>(x.U) + so);
            x.Y = reinterpret_cast<double *>(reinterpret_cast<float *>(x.Y) + so);
        } else {
            x.U += so;
            x.Y += so;
        }
        if (x.LAM) x.LAM += so;
        if (x.U0) x.U0 += so;
        if (x.pay_mis) x.pay_mis += o;
        if (x.R1) x.R1 += so;
        if (x.C2) x.C2 += so;
        if (x.rs_tab) x.rs_tab += (size_t)o * L.nrows * 128;
        x.scoef += (size_t)o * pl.n_scoef; x.b2row += (size_t)o * pl.n_b2row; x.rowc += (size_t)o * pl.n_rowc;
        x.pb += (size_t)o * pl.n_pb; x.rinv += (size_t)o * pl.n_rinv; x.ipar += o;
        return x;
    };
    const HadiSweepArgs a_all = a, av_all = av;
    bool forked = false;
    auto enqueue_body = [&](hipStream_t q0) -> int {
      const int n_first = d.debug ? d.debug_step : 1, n_last = d.debug ? d.debug_step : d.Nmax;
      for (int sb = 0; sb < nsub; sb++) {  // one sub-batch after the other (per stream), each through its whole time loop
        if (two_streams && sb == fork_before) {  // fork: the second stream starts behind everything enqueued so far
            HIP_TRY(c, hipEventRecord(c->fork_ev, q0));
            HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->fork_ev, 0));
            forked = true;
        }
        hipStream_t q = (two_streams && subs[sb].lane) ? c->stream2 : q0;
        const int o = subs[sb].off, nsb = subs[sb].cnt;
        const HadiPlan &pl = subs[sb].pl;  // (shadows the whole-batch plan: launch geometry of THIS sub-batch)
        const size_t so = (size_t)o * L.inst_stride;
        const HadiSweepArgs a = shift(a_all, o, nsb, pl), av = shift(av_all, o, nsb, pl);
        const size_t tot = (size_t)L.inst_stride * nsb, st = tot * sizeof(double);  // (shadow the whole-batch sizes)
        double *const Ub = ptr<double>(c->U) + so, *const LAMb = american ? ptr<double>(c->LAM) + so : nullptr;
        double *const U0b = american ? ptr<double>(c->U0) + so : nullptr, *const UTb = dividend ? ptr<double>(c->UT) + so : nullptr;
        const int ev0 = 4 * sb * d.Nmax;  // profiling events of this sub-batch
        if (pair_tab && pl.use_strip) {  // paired strips: the pairs' coupling column, once per solve (hadi_strip_step, RSTAB)
            HadiSweepArgs at = a;
            at.U = Ub;  // (any packed fp64 array: the table depends on the matrix only)
            // (its own LDS size: the fp64 ring of 4 pairs x 3 slots, whatever the state precision of the sweep -- hadi_plan.h)
            const size_t sm = (size_t)4 * HADI_STRIP_NS(8, 2, 8) * L.rowp * sizeof(double) + ((size_t)4 * 64 * 8 * 2 + (size_t)4 * 16) * sizeof(double);
            hipLaunchKernelGGL((hadi_pass_a_strip<8, 0, double, 2, 3>), dim3(pl.grid_as), dim3(512), sm, q, at, 1);
        }
        for (int nstep = n_first; nstep <= n_last; nstep++) {
            // P representation: the first step (the caller's initial U need not dominate the payoff) and dividend steps
            // (the jump acts on U alone) run on the explicit (U, lambda_bar) pair, converted on the way in and out
            const bool xstep = amp && (nstep == 1 || (have_div && div_step[nstep]));
            if (xstep && nstep > 1)
                hipLaunchKernelGGL(hadi_am_materialise_kernel, dim3(grid1d(tot)), dim3(256), 0, q, L, nsb, a.ipar, U0b, Ub, LAMb, pl.pos_m1);
            if (have_div && div_step[nstep]) {  // device_solver.hpp:426-517: U_temp <- U, U <- interpolated jump
                if (f32)  // fp32 state: the jump works on the fp64 packed array -- widen, jump, round again (<= num_dividends steps)
                    hipLaunchKernelGGL(hadi_widen_kernel, dim3(grid1d(tot)), dim3(256), 0, q, L, reinterpret_cast<const float *>(a.U), Ub, tot);
                HIP_TRY(c, hipMemcpyAsync(UTb, Ub, st, hipMemcpyDeviceToDevice, q));
                const size_t npts = (size_t)nsb * L.nrows * (L.m1 + 1);
                hipLaunchKernelGGL(hadi_dividend_kernel, dim3(grid1d(npts)), dim3(256), 0, q, L, nsb, a.ipar,
                                   d.d_vec_s + (size_t)o * (L.m1 + 1), UTb, Ub, ptr<int>(c->div_flag) + (size_t)o * flag_stride,
                                   flag_stride, nstep, ptr<double>(c->div_amt), ptr<double>(c->div_pct));
                if (f32)
                    hipLaunchKernelGGL(hadi_narrow_kernel, dim3(grid1d(tot)), dim3(256), 0, q, L, Ub, reinterpret_cast<float *>(a.U), tot);
            }
            if (prof) HIP_TRY(c, hipEventRecord(c->kev[ev0 + 4 * (nstep - 1) + 0], q));
            const PassEnv env{pl, L, nsb, q, nstep, american, amp, xstep, f32, c->col_prefetch, c->cs_strips};
            auto row_pass = [&](const HadiSweepArgs &ar, int mode) { launch_row_pass(env, ar, mode); };
            auto col_pass = [&](const HadiSweepArgs &ar) { launch_col_pass(env, ar); };
            if (d.debug == 2) {  // diagnostics: one column solve of the packed input (moved to Y), nothing else
                HIP_TRY(c, hipMemcpyAsync(a.Y, a.U, f32 ? st / 2 : st, hipMemcpyDeviceToDevice, q));
                col_pass(a);
                break;
            }
            row_pass(a, cs ? 1 : 0);
            if (d.debug == 1) break;  // diagnostics: Y now holds the right-hand side of the A2 solve
            if (prof) {
                HIP_TRY(c, hipEventRecord(c->kev[ev0 + 4 * (nstep - 1) + 1], q));
                HIP_TRY(c, hipEventRecord(c->kev[ev0 + 4 * (nstep - 1) + 2], q));
            }
            col_pass(cs ? av : a);
            if (prof) HIP_TRY(c, hipEventRecord(c->kev[ev0 + 4 * (nstep - 1) + 3], q));
            if (cs) {  // corrector (profiling events cover the predictor's two passes only)
                row_pass(av, 2);
                col_pass(a);
            }
            if (xstep)
                hipLaunchKernelGGL(hadi_am_dematerialise_kernel, dim3(grid1d(tot)), dim3(256), 0, q, L, nsb, a.ipar, U0b, Ub, LAMb);
        }
        if (amp)  // explicit U and lambda_bar for the outputs
            hipLaunchKernelGGL(hadi_am_materialise_kernel, dim3(grid1d(tot)), dim3(256), 0, q, L, nsb, a.ipar, U0b, Ub, LAMb, pl.pos_m1);
      }
      return HADI_OK;
    };
    // Fork / join around the body.  The join is enqueued even when the body failed half way: a second stream left un-joined
    // would make hipStreamEndCapture fail ("unjoined work") and stay in capture mode for the handle's next call.
    auto enqueue_loop = [&](hipStream_t q0) -> int {
      forked = false;
      const int rcb = enqueue_body(q0);
      if (forked) {  // join
          const hipError_t e1 = hipEventRecord(c->join_ev, c->stream2);
          const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(q0, c->join_ev, 0) : e1;
          if (!rcb && e2 != hipSuccess) return fail(c, HADI_ERR_HIP, "joining the second stream failed: %s", hipGetErrorString(e2));
      }
      return rcb;
    };

    // ---- small grids: the whole instance fits in LDS -> one launch runs the entire time loop ----------
    const size_t smem_small = american ? pl.smem_small_am : pl.smem_small_eu;
    // European / dividend sweeps: one wavefront per instance with sequential line solves (hadi_small_seq_kernel) issues about
    // half the instructions per instance and step but runs them on ONE wavefront -- ahead once there are more instances than
    // CUs (50x25, 40 steps, ms block kernel / this one: 256 instances 0.49 / 0.55, 320: 0.67 / 0.60, 512: 0.71 / 0.63, 768: 0.95 /
    // 0.80; 3000 x 50 steps: 3.75 / 2.13), behind below that (a single instance: 10 against 12 us per step).
    // "small_seq" = 1 forces it, 0 forbids it, -1 (default) picks by batch size.
    const bool seq = takes_small_path && !american && (c->small_seq > 0 || (c->small_seq < 0 && d.n > c->cu_count));
    const size_t smem_seq = (size_t)hadi_small_seq_layout(L.m1, L.nrows).total * sizeof(double);
    // ... and two instances per wavefront for batches of more than 2 and at most 4.5 instances per CU: a wavefront then retires
    // two instances' steps in 1.15x the time of one, but the launch has half the wavefronts -- below 2 per CU the instances are
    // better spread over the CUs, at the 6 per CU that the LDS holds either way the halved instruction count and the halved
    // latency hiding cancel (50x25 x 200 steps, ms: 768 instances 3.20 -> 2.76, 1024: 3.49 -> 2.78, 1536: 3.53 -> 3.83, 3072:
    // 6.68 -> 6.98).  "small_pairs" = 1 forces it, 0 forbids it, -1 (default) picks by batch size.  Needs nrows <= 32.
    const bool seq2 = seq && L.nrows <= 32 && 2 * smem_seq <= (size_t)160 * 1024 &&
                      (c->small_pairs > 0 || (c->small_pairs < 0 && d.n > 2 * c->cu_count && 2 * d.n <= 9 * c->cu_count));
    if (takes_small_path) {
        {
            char buf[192];
            if (seq2)
                std::snprintf(buf, sizeof buf, "hadi_small_seq2_kernel<%d>: whole time loop in one launch, two instances per wavefront, lines solved sequentially in LDS (2 x %zu B)", L.B, smem_seq);
            else if (seq)
                std::snprintf(buf, sizeof buf, "hadi_small_seq_kernel<%d>: whole time loop in one launch, one wavefront per instance, lines solved sequentially in LDS (%zu B)", L.B, smem_seq);
            else
                std::snprintf(buf, sizeof buf, "hadi_small_kernel<%d,%d,%s>: whole time loop in one launch, instance resident in LDS (%zu B)", L.B,
                              (c->tune.small_waves ? c->tune.small_waves : (d.n <= 2 * c->cu_count ? 8 : 4)) == 8 ? 8 : 4, american ? "AM" : "EU", smem_small);
            c->last_path = buf;
        }
        HadiSmallArgs sm;
        sm.div_flag = nullptr; sm.div_amounts = nullptr; sm.div_pcts = nullptr; sm.vec_s = d.d_vec_s; sm.Nmax = d.Nmax;
        sm.order = nullptr;
        if (!d.uniform_steps) {  // longest-processing-time-first dispatch order
            std::vector<int> order(d.n);
            for (int k = 0; k < d.n; k++) order[k] = k;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
                return d.par8[(size_t)x * 8 + 5] > d.par8[(size_t)y * 8 + 5];
            });
            if ((rc = ensure(c, c->order, sizeof(int) * n))) return rc;
            if ((rc = stage_to_device(c, c->order.p, order.data(), sizeof(int) * n))) return rc;
            sm.order = ptr<int>(c->order);
        }
        sm.flag_stride = flag_stride;
        if (have_div) {
            sm.div_flag = ptr<int>(c->div_flag); sm.div_amounts = ptr<double>(c->div_amt); sm.div_pcts = ptr<double>(c->div_pct);
        }
        HIP_TRY(c, hipEventRecord(c->ev[1], s));
        // wavefronts per instance: 4 when the batch fills the GPU (throughput), 8 for small batches (latency of the
        // dependent per-step phases; more waves share the rows of the row pass)
        // (measured, 50x25 grid: 1 instance x 100 steps 1.27 -> 1.04 ms with 8; 3000 instances x 50 steps 4.19 -> 4.58 ms)
        const int sw = c->tune.small_waves ? c->tune.small_waves : (d.n <= 2 * c->cu_count ? 8 : 4);
        if (seq2) {
            if (L.B == 1) hipLaunchKernelGGL((hadi_small_seq2_kernel<1>), dim3((d.n + 1) / 2), dim3(64), 2 * smem_seq, s, a, sm);
            else hipLaunchKernelGGL((hadi_small_seq2_kernel<2>), dim3((d.n + 1) / 2), dim3(64), 2 * smem_seq, s, a, sm);
        } else if (seq) {
            if (L.B == 1) hipLaunchKernelGGL((hadi_small_seq_kernel<1>), dim3(d.n), dim3(64), smem_seq, s, a, sm);
            else hipLaunchKernelGGL((hadi_small_seq_kernel<2>), dim3(d.n), dim3(64), smem_seq, s, a, sm);
        } else if (sw == 8) {
            if (L.B == 1) {
                if (american) hipLaunchKernelGGL((hadi_small_kernel<1, 8, true>), dim3(d.n), dim3(512), smem_small, s, a, sm);
                else hipLaunchKernelGGL((hadi_small_kernel<1, 8, false>), dim3(d.n), dim3(512), smem_small, s, a, sm);
            } else {
                if (american) hipLaunchKernelGGL((hadi_small_kernel<2, 8, true>), dim3(d.n), dim3(512), smem_small, s, a, sm);
                else hipLaunchKernelGGL((hadi_small_kernel<2, 8, false>), dim3(d.n), dim3(512), smem_small, s, a, sm);
            }
        } else if (L.B == 1) {
            if (american) hipLaunchKernelGGL((hadi_small_kernel<1, 4, true>), dim3(d.n), dim3(256), smem_small, s, a, sm);
            else hipLaunchKernelGGL((hadi_small_kernel<1, 4, false>), dim3(d.n), dim3(256), smem_small, s, a, sm);
        } else {
            if (american) hipLaunchKernelGGL((hadi_small_kernel<2, 4, true>), dim3(d.n), dim3(256), smem_small, s, a, sm);
            else hipLaunchKernelGGL((hadi_small_kernel<2, 4, false>), dim3(d.n), dim3(256), smem_small, s, a, sm);
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->ev[2], s));
        return HADI_OK;
    }

    c->last_path = describe_streaming_path(c, pl, bp, american, amp, cs, f32);
    c->last_nsub = nsub;
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    // ---- instance-resident launch: up to 8 large European instances, one per XCD, whole time loop in one kernel ----------
    // (hadi_team_kernel; the reference runs every instance's time loop inside one kernel, device_solver.hpp:83-88,226-265).
    // Chosen automatically for batches of up to 8 instances on the full 256-CU device; any failure of the team protocol is
    // recorded by the kernel, checked here, and the batch is solved again on the streaming path below.
    const bool team_shape = d.n <= 8 && L.G == 1 && (L.B == 8 || L.B == 4) && L.P <= 8 && !seq_shape && (d.variant == HADI_EU || d.variant == HADI_DIV) && !cs && !f32 &&
                            !d.debug && !prof && d.theta > 0.0 && d.r_d != d.r_f && c->cu_count == 256;
    // (a caller who pins the streaming kernels' geometry -- hadi_set_tuning "strip", "row_tile", "col_groups", "strip_blocks" --
    // gets those kernels)
    const bool pinned = c->tune.strip >= 0 || c->tune.row_tile > 0 || c->tune.col_groups > 0 || c->tune.strip_blocks > 0;
    if (team_shape && (c->team_launch > 0 || (c->team_launch < 0 && !c->team_failed && !pinned))) {
        if ((rc = ensure(c, c->team, 512 * sizeof(int)))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->team.p, 0, 512 * sizeof(int), s));
        HadiTeamArgs ta;
        ta.form = ptr<int>(c->team); ta.bar = ptr<int>(c->team) + 64; ta.nb = c->cu_count / 8; ta.N = d.Nmax;
        ta.div_flag = have_div ? ptr<int>(c->div_flag) : nullptr; ta.flag_stride = flag_stride;
        ta.div_amounts = have_div ? ptr<double>(c->div_amt) : nullptr; ta.div_pcts = have_div ? ptr<double>(c->div_pct) : nullptr;
        ta.vec_s = d.d_vec_s;
        ta.stamps = reinterpret_cast<unsigned long long *>(ptr<int>(c->team) + 384);
        const size_t smem = ((size_t)4 * 64 * L.B + hadi_pb_mf_doubles(L.P) + (size_t)L.P * HADI_LC * HADI_PBW +
                             (have_div ? (size_t)(L.m1 + 2) + (size_t)8 * L.rowp : 0)) * sizeof(double) + 64;
        if (L.B == 8) hipLaunchKernelGGL((hadi_team_kernel<8>), dim3(c->cu_count), dim3(512), smem, s, a, ta);
        else hipLaunchKernelGGL((hadi_team_kernel<4>), dim3(c->cu_count), dim3(512), smem, s, a, ta);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(s));
        const int deverr = __atomic_exchange_n(c->err_host, 0, __ATOMIC_ACQ_REL);
        if (!deverr) {
            char buf[200];
            std::snprintf(buf, sizeof buf, "hadi_team_kernel<%d>: whole time loop in one launch, every instance resident in one XCD's L2 (teams of %d blocks)",
                          L.B, ta.nb);
            c->last_path = buf;
            HIP_TRY(c, hipEventRecord(c->ev[2], s));
            return HADI_OK;
        }
        if (deverr & ~HADI_DEVERR_TEAM) __atomic_fetch_or(c->err_host, deverr & ~HADI_DEVERR_TEAM, __ATOMIC_RELAXED);  // (not ours: keep it for finish_timing)
        c->team_failed = 1;
        // start again from the caller's initial condition on the streaming path
        hipLaunchKernelGGL(hadi_pack_kernel, dim3(grid1d(tot)), dim3(256), 0, s, L, d.n, d.n_src, d.d_natU, ptr<double>(c->U));
        HIP_TRY(c, hipMemsetAsync(c->Y.p, 0, st, s));
        c->last_path += " (after a failed instance-resident launch)";
    }
    // Small batches are launch-bound (2*N dependent launches of a few microseconds each): replay the loop
    // from a cached hipGraph.  Every kernel argument is baked into the nodes, so the key is everything they
    // depend on; the library's own buffers are stable between calls.
    const bool graphable = c->use_graph && !prof && !d.debug && (long long)d.n * L.inst_stride <= ((long long)c->graph_max_melems << 20);
    if (graphable) {
        std::string key;
        auto put = [&](const void *p_, size_t nbytes) { key.append(static_cast<const char *>(p_), nbytes); };
        {  // field by field: struct padding is not initialised
            const void *ptrs[] = {a.pay_mis, a.U, a.Y, a.LAM, a.U0, a.scoef, a.b2row, a.rowc, a.pb, a.rinv, a.ipar, a.R1, a.C2, av.U, a.err};
            const int ints[] = {a.debug, a.L.m1, a.L.m2, a.L.B, a.L.G, a.L.P, a.n_inst, a.R, a.ntiles, a.ctiles, a.btpw, a.bgroups,
                                a.american, a.pos_m1, d.scheme, d.prec, (int)amp, (int)two_streams, nsub, pl.row_seq, pl.col_seq, pl.use_pairs, pl.use_strip, pl.RS, pl.sblocks, pl.grid_as, pl.grid_a, pl.grid_b, pl.block_b, pl.W, (int)pl.smem_a, (int)pl.smem_b};
            put(ptrs, sizeof(ptrs));
            put(ints, sizeof(ints));
        }
        for (const auto &sbt : subs) {  // the launch geometry of EVERY sub-batch is baked into the nodes (unequal halves on two
                                        // streams, a tuning change that flips only the second sub-batch's plan)
            const HadiPlan &q = sbt.pl;
            const int geo[] = {c->col_prefetch, c->cs_strips, c->tile_il, sbt.lane, fork_before, sbt.off, sbt.cnt, q.R, q.ntiles, q.grid_a, (int)q.smem_a, q.use_strip, q.use_pairs, q.RS, q.sblocks, q.grid_as,
                               (int)q.smem_as, q.ctiles, q.btpw, q.bgroups, q.grid_b, q.block_b, (int)q.smem_b, q.row_seq, q.col_seq, q.W, q.NG, q.PD};
            put(geo, sizeof(geo));
        }
        put(&d.Nmax, sizeof(int)); put(&d.dt0, sizeof(double));
        put(&d.variant, sizeof(int)); put(&d.d_vec_s, sizeof(void *));
        void *ut = c->UT.p; put(&ut, sizeof(ut));
        if (have_div) {  // amounts / percentages / per-instance tables are re-uploaded every call; the node list
                         // only depends on which steps carry a dividend launch
            put(&flag_stride, sizeof(int));
            put(div_step.data(), div_step.size());
            void *fl = c->div_flag.p, *am = c->div_amt.p, *pc = c->div_pct.p;
            put(&fl, sizeof(fl)); put(&am, sizeof(am)); put(&pc, sizeof(pc));
        }
        Ctx::GraphEntry *hit = nullptr;
        for (auto &g : c->graphs)
            if (g.key == key) { hit = &g; break; }
        if (!hit) {
            if (c->graphs.size() >= 8) {  // evict the least recently used entry
                size_t lru = 0;
                for (size_t k = 1; k < c->graphs.size(); k++)
                    if (c->graphs[k].stamp < c->graphs[lru].stamp) lru = k;
                (void)hipGraphExecDestroy(c->graphs[lru].exec);
                (void)hipGraphDestroy(c->graphs[lru].graph);
                c->graphs.erase(c->graphs.begin() + lru);
            }
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            HIP_TRY(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const int rcl = enqueue_loop(s);
            hipError_t ec = hipStreamEndCapture(s, &graph);
            if (rcl || ec != hipSuccess) {  // nothing half-built survives: the captured graph is dropped
                if (graph) (void)hipGraphDestroy(graph);
                if (rcl) return rcl;
            }
            HIP_TRY(c, ec);
            HIP_TRY(c, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            c->graphs.push_back(Ctx::GraphEntry{key, graph, exec, 0});
            hit = &c->graphs.back();
        }
        hit->stamp = ++c->graph_clock;
        HIP_TRY(c, hipGraphLaunch(hit->exec, s));
    } else {
        const int rcl = enqueue_loop(s);
        if (rcl) return rcl;
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    if (f32)  // back to the fp64 packed array the unpack / price-pick kernels read
        hipLaunchKernelGGL(hadi_widen_kernel, dim3(grid1d(tot)), dim3(256), 0, s, L, ptr<float>(c->Uf), ptr<double>(c->U), tot);
    return HADI_OK;
}

int finish_timing(Ctx *c, const SweepDesc &d, const HadiPlan &pl) {
    hipStream_t s = c->stream;
    HIP_TRY(c, hipEventRecord(c->ev[3], s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->pin_dirty = 0;
    // the sticky device error word (see Ctx::err_host): whatever a kernel of this call reported is visible now
    const int deverr = __atomic_exchange_n(c->err_host, 0, __ATOMIC_ACQ_REL);
    float ms = 0;
    hadi_timing &t = c->timing;
    t = hadi_timing{};
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); t.setup_ms = ms;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); t.sweep_ms = ms;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); t.finish_ms = ms;
    long long ps = 0;
    const long long m = (long long)(d.m1 + 1) * (d.m2 + 1);
    for (int k = 0; k < d.n; k++) ps += m * (long long)d.par8[(size_t)k * 8 + 5];
    t.point_steps = ps;
    if (c->profiling && !d.debug && (int)c->kev.size() >= 4 * d.Nmax * c->last_nsub) {
        for (int k = 0; k < d.Nmax * c->last_nsub; k++) {
            HIP_TRY(c, hipEventElapsedTime(&ms, c->kev[4 * k], c->kev[4 * k + 1])); t.pass_a_ms += ms;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->kev[4 * k + 2], c->kev[4 * k + 3])); t.pass_b_ms += ms;
        }
        t.pass_a_launches = (long long)d.Nmax * c->last_nsub;
        t.pass_b_launches = (long long)d.Nmax * c->last_nsub;
    }
    (void)pl;
    if (deverr)
        return fail(c, HADI_ERR_INTERNAL, "device-side failure 0x%x during the sweep%s: the results of this call are invalid", deverr,
                    (deverr & HADI_DEVERR_RENDEZVOUS) ? " (a pair rendezvous of a two-wavefront row ran out of polls)" : "");
    return HADI_OK;
}

// Brings an array argument to the device (staging host memory into `buf`).
int to_device(Ctx *c, int memspace, const double *src, size_t count, DevBuf &buf, const double **out) {
    if (!src) { *out = nullptr; return HADI_OK; }
    if (memspace == HADI_MEM_DEVICE) { *out = src; return HADI_OK; }
    int rc = ensure(c, buf, count * sizeof(double));
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(buf.p, src, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    *out = ptr<double>(buf);
    return HADI_OK;
}

int from_device(Ctx *c, int memspace, double *dst, const double *dsrc, size_t count) {
    HIP_TRY(c, hipMemcpyAsync(dst, dsrc, count * sizeof(double),
                              memspace == HADI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                              c->stream));
    return HADI_OK;
}

int check_problem(Ctx *c, const hadi_problem *p, bool need_U, bool need_vgrid) {
    if (!c) return HADI_ERR_INVALID;
    if (!p) return fail(c, HADI_ERR_INVALID, "problem is NULL");
    if (p->n_instances < 1) return fail(c, HADI_ERR_INVALID, "n_instances must be >= 1");
    if (p->variant < HADI_EU || p->variant > HADI_AM_DIV) return fail(c, HADI_ERR_INVALID, "bad variant %d", p->variant);
    if (p->memspace != HADI_MEM_HOST && p->memspace != HADI_MEM_DEVICE)
        return fail(c, HADI_ERR_INVALID, "bad memspace %d", p->memspace);
    if (!p->vec_s || !p->delta_s) return fail(c, HADI_ERR_INVALID, "vec_s / delta_s missing");
    if (need_vgrid && (!p->vec_v || !p->delta_v)) return fail(c, HADI_ERR_INVALID, "vec_v / delta_v missing");
    if (need_U && !p->U) return fail(c, HADI_ERR_INVALID, "U missing");
    if (!p->N_i && p->N < 1) return fail(c, HADI_ERR_INVALID, "N must be >= 1");
    if (!p->delta_t_i && !(p->delta_t > 0 && std::isfinite(p->delta_t))) return fail(c, HADI_ERR_INVALID, "delta_t must be > 0");
    for (int k = 0; k < p->n_instances; k++) {  // per-instance overrides: every entry, not only the shared scalars
        if (p->N_i && p->N_i[k] < 1) return fail(c, HADI_ERR_INVALID, "N_i[%d] = %d must be >= 1", k, p->N_i[k]);
        if (p->delta_t_i && !(p->delta_t_i[k] > 0 && std::isfinite(p->delta_t_i[k])))
            return fail(c, HADI_ERR_INVALID, "delta_t_i[%d] = %g must be > 0 and finite", k, p->delta_t_i[k]);
    }
    if (!(p->theta >= 0 && std::isfinite(p->theta))) return fail(c, HADI_ERR_INVALID, "theta must be >= 0");
    if (p->option_type != HADI_CALL && p->option_type != HADI_PUT) return fail(c, HADI_ERR_INVALID, "bad option_type %d", p->option_type);
    if (p->option_type == HADI_PUT) {
        if (!p->strike_i) return fail(c, HADI_ERR_INVALID, "option_type = HADI_PUT needs strike_i (boundary value K e^{-r_d t})");
        for (int k = 0; k < p->n_instances; k++)
            if (!(p->strike_i[k] > 0 && std::isfinite(p->strike_i[k]))) return fail(c, HADI_ERR_INVALID, "strike_i[%d] must be > 0", k);
        if (p->scheme != HADI_SCHEME_DOUGLAS) return fail(c, HADI_ERR_UNSUPPORTED, "put boundary data are available for Douglas steps only");
    }
    if (p->V_0_i && need_vgrid) return fail(c, HADI_ERR_INVALID, "V_0_i applies to hadi_compute_base_prices* / hadi_compute_jacobian* only");
    {  // grid shape, before anything is staged
        HadiPlan tmp;
        if (p->m1 < 2 || p->m2 < 3 ||
            hadi_make_plan(p->m1, p->m2, p->n_instances, 8 * c->cu_count, &tmp, c->tune, p->state_precision == HADI_STATE_FP32 ? 4 : 8))
            return fail(c, HADI_ERR_UNSUPPORTED, "grid %dx%d not supported (need m1 >= 2, m2 >= 3 and (m1 + 16)(m2 + 1) < 2^28)", p->m1, p->m2);
    }
    const bool dividend = p->variant == HADI_DIV || p->variant == HADI_AM_DIV;
    if (dividend && p->num_dividends > 0 && (!p->dividend_dates || !p->dividend_amounts || !p->dividend_percentages))
        return fail(c, HADI_ERR_INVALID, "dividend arrays missing");
    if (p->num_dividends < 0) return fail(c, HADI_ERR_INVALID, "num_dividends < 0");
    if (p->scheme != HADI_SCHEME_DOUGLAS && p->scheme != HADI_SCHEME_CRAIG_SNEYD)
        return fail(c, HADI_ERR_INVALID, "bad scheme %d", p->scheme);
    if (p->scheme == HADI_SCHEME_CRAIG_SNEYD && p->variant != HADI_EU)
        return fail(c, HADI_ERR_UNSUPPORTED, "Craig-Sneyd is available for the European variant only (as in the reference)");
    if (p->state_precision != HADI_STATE_FP64 && p->state_precision != HADI_STATE_FP32)
        return fail(c, HADI_ERR_INVALID, "bad state_precision %d", p->state_precision);
    if (p->state_precision == HADI_STATE_FP32 &&
        ((p->variant != HADI_EU && p->variant != HADI_DIV) || p->scheme != HADI_SCHEME_DOUGLAS))
        return fail(c, HADI_ERR_UNSUPPORTED, "the fp32-state sweep covers European Douglas steps (with or without dividends) only");
    return HADI_OK;
}

// Fills the per-instance parameter rows for `groups` copies of the caller's batch.
void fill_par(const hadi_problem *p, SweepDesc &d, int groups) {
    const int n0 = p->n_instances;
    d.par8.assign((size_t)n0 * groups * 8, 0.0);
    d.Nmax = 0;
    d.uniform_steps = !(p->N_i || p->delta_t_i);
    d.dt0 = p->delta_t;
    for (int g = 0; g < groups; g++)
        for (int k = 0; k < n0; k++) {
            double *r = &d.par8[((size_t)g * n0 + k) * 8];
            r[0] = p->rho_i ? p->rho_i[k] : p->rho;
            r[1] = p->sigma_i ? p->sigma_i[k] : p->sigma;
            r[2] = p->kappa_i ? p->kappa_i[k] : p->kappa;
            r[3] = p->eta_i ? p->eta_i[k] : p->eta;
            r[4] = p->delta_t_i ? p->delta_t_i[k] : p->delta_t;
            const int N = p->N_i ? p->N_i[k] : p->N;
            r[5] = (double)N;
            r[6] = (p->option_type == HADI_PUT) ? p->strike_i[k] : 0.0;
            r[7] = (p->option_type == HADI_PUT) ? 1.0 : 0.0;
            d.Nmax = std::max(d.Nmax, N);
        }
}

void fill_common(const hadi_problem *p, SweepDesc &d) {
    d.m1 = p->m1; d.m2 = p->m2; d.variant = p->variant; d.scheme = p->scheme; d.prec = p->state_precision;
    d.theta = p->theta; d.r_d = p->r_d; d.r_f = p->r_f;
    const bool dividend = p->variant == HADI_DIV || p->variant == HADI_AM_DIV;
    d.num_div = dividend ? p->num_dividends : 0;
    d.div_dates = p->dividend_dates; d.div_amounts = p->dividend_amounts; d.div_pcts = p->dividend_percentages;
}

// v-grids of all `n` instances rebuilt ON THE DEVICE, each for its own V_0 (v0i: host vector of n values): what every
// team of the reference does in-kernel (rebuild_variance_views, grid_pod.hpp:25-73; call sites use V = 5, d = 5/500,
// jacobian_computation.cpp:253).  Results in c->g_v / c->g_dv; c->v0_i keeps the per-instance V_0 for the price pick.
int rebuild_v_device(Ctx *c, int n, int m2, const std::vector<double> &v0i) {
    int rc;
    if ((rc = ensure(c, c->v0_i, (size_t)n * 8)) || (rc = ensure(c, c->g_v, (size_t)n * (m2 + 1) * 8)) ||
        (rc = ensure(c, c->g_dv, (size_t)n * m2 * 8)))
        return rc;
    if ((rc = stage_to_device(c, c->v0_i.p, v0i.data(), (size_t)n * 8))) return rc;
    hipLaunchKernelGGL(hadi_rebuild_variance_kernel, dim3(n), dim3(256), (size_t)2 * (m2 + 1) * sizeof(double), c->stream, m2, n,
                       ptr<double>(c->v0_i), 5.0, 5.0 / 500, ptr<double>(c->g_v), ptr<double>(c->g_dv));
    HIP_TRY(c, hipGetLastError());
    return HADI_OK;
}

// Shared driver of hadi_DO_timestepping / hadi_parallel_DO_solve / hadi_compute_base_prices* and the diagnostics
// (debug != 0: p->U is input only, the pass's result goes to debug_out).
int solve_common(Ctx *c, const hadi_problem *p, bool rebuild_v, bool pick, double S_0, double V_0, double *prices_out,
                 int debug = 0, int debug_step = 1, double *debug_out = nullptr) {
    int rc = check_problem(c, p, true, !rebuild_v);
    if (rc) return rc;
    if (debug && !debug_out) return fail(c, HADI_ERR_INVALID, "output array missing");
    if (debug && (p->scheme != HADI_SCHEME_DOUGLAS || p->state_precision != HADI_STATE_FP64))
        return fail(c, HADI_ERR_UNSUPPORTED, "diagnostics cover fp64 Douglas steps");
    if (debug == 2 && p->variant != HADI_EU && p->variant != HADI_DIV)
        return fail(c, HADI_ERR_UNSUPPORTED, "hadi_debug_col_solve is the plain A2 solve (no projection): European variants only");
    DeviceGuard guard(c->device);
    pin_rewind(c);
    const int n = p->n_instances, m1 = p->m1, m2 = p->m2;
    const size_t m = (size_t)(m1 + 1) * (m2 + 1);
    SweepDesc d;
    fill_common(p, d);
    d.n = n; d.n_src = n;
    d.debug = debug; d.debug_step = debug_step;
    fill_par(p, d, 1);
    if (debug == 1 && (debug_step < 1 || debug_step > d.Nmax)) return fail(c, HADI_ERR_INVALID, "step %d outside 1..%d", debug_step, d.Nmax);
    if ((rc = to_device(c, p->memspace, p->vec_s, (size_t)n * (m1 + 1), c->g_s, &d.d_vec_s))) return rc;
    if ((rc = to_device(c, p->memspace, p->delta_s, (size_t)n * m1, c->g_ds, &d.d_delta_s))) return rc;
    const bool per_inst_v0 = rebuild_v && c->device_vgrid;
    if (rebuild_v && !c->device_vgrid && p->V_0_i)
        return fail(c, HADI_ERR_INVALID, "V_0_i needs the device v-grid rebuild (hadi_set_tuning \"device_vgrid\", 1)");
    if (per_inst_v0) {
        std::vector<double> v0i(n);
        for (int k = 0; k < n; k++) v0i[k] = p->V_0_i ? p->V_0_i[k] : V_0;
        if ((rc = rebuild_v_device(c, n, m2, v0i))) return rc;
        d.d_vec_v = ptr<double>(c->g_v);
        d.d_delta_v = ptr<double>(c->g_dv);
    } else if (rebuild_v) {
        // host build (glibc sinh/asinh, bit-identical to the reference's host-side Grid), broadcast to every instance
        std::vector<double> hv(m2 + 1), hdv(m2);
        build_v(m2, V_0, 5.0, 5.0 / 500, hv.data(), hdv.data());
        if ((rc = ensure(c, c->src_v, (m2 + 1) * 8)) || (rc = ensure(c, c->src_dv, m2 * 8))) return rc;
        if ((rc = ensure(c, c->g_v, (size_t)n * (m2 + 1) * 8)) || (rc = ensure(c, c->g_dv, (size_t)n * m2 * 8))) return rc;
        if ((rc = stage_to_device(c, c->src_v.p, hv.data(), (size_t)(m2 + 1) * 8)) || (rc = stage_to_device(c, c->src_dv.p, hdv.data(), (size_t)m2 * 8))) return rc;
        hipLaunchKernelGGL(hadi_bcast_rows_kernel, dim3(grid1d((size_t)n * (m2 + 1))), dim3(256), 0, c->stream, m2 + 1, n,
                           ptr<double>(c->src_v), (const int *)nullptr, ptr<double>(c->g_v));
        hipLaunchKernelGGL(hadi_bcast_rows_kernel, dim3(grid1d((size_t)n * m2)), dim3(256), 0, c->stream, m2, n,
                           ptr<double>(c->src_dv), (const int *)nullptr, ptr<double>(c->g_dv));
        d.d_vec_v = ptr<double>(c->g_v);
        d.d_delta_v = ptr<double>(c->g_dv);
    } else {
        if ((rc = to_device(c, p->memspace, p->vec_v, (size_t)n * (m2 + 1), c->g_v, &d.d_vec_v))) return rc;
        if ((rc = to_device(c, p->memspace, p->delta_v, (size_t)n * m2, c->g_dv, &d.d_delta_v))) return rc;
    }
    if ((rc = to_device(c, p->memspace, p->U, n * m, c->natU, &d.d_natU))) return rc;
    if ((rc = to_device(c, p->memspace, p->U_0, n * m, c->natU0, &d.d_natU0))) return rc;

    HadiPlan pl;
    if ((rc = run_sweep(c, d, pl))) return rc;

    // solution back to the caller's U (natural layout); diagnostics: the pass's result to debug_out, U untouched
    double *const host_dst = debug ? debug_out : p->U;
    double *d_out;
    if (p->memspace == HADI_MEM_DEVICE) d_out = host_dst;
    else {
        if ((rc = ensure(c, c->natOut, n * m * 8))) return rc;
        d_out = ptr<double>(c->natOut);
    }
    hipLaunchKernelGGL(hadi_unpack_kernel, dim3(grid1d(n * m)), dim3(256), 0, c->stream, pl.L, n,
                       debug == 1 ? ptr<double>(c->Y) : ptr<double>(c->U), d_out);
    if (p->memspace == HADI_MEM_HOST && (rc = from_device(c, p->memspace, host_dst, d_out, n * m))) return rc;
    const bool american = (p->variant == HADI_AM || p->variant == HADI_AM_DIV) && !debug;
    if (american && p->lambda_bar) {
        double *d_l;
        if (p->memspace == HADI_MEM_DEVICE) d_l = p->lambda_bar;
        else {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            d_l = ptr<double>(c->natOut);
        }
        hipLaunchKernelGGL(hadi_unpack_kernel, dim3(grid1d(n * m)), dim3(256), 0, c->stream, pl.L, n, ptr<double>(c->LAM), d_l);
        if (p->memspace == HADI_MEM_HOST && (rc = from_device(c, p->memspace, p->lambda_bar, d_l, n * m))) return rc;
    }
    std::vector<int> hstatus;
    if (pick) {
        if ((rc = ensure(c, c->prices, n * 8)) || (rc = ensure(c, c->status, n * sizeof(int)))) return rc;
        hipLaunchKernelGGL(hadi_pick_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, pl.L, n, d.d_vec_s, d.d_vec_v,
                           ptr<double>(c->U), S_0, per_inst_v0 ? ptr<double>(c->v0_i) : (const double *)nullptr, V_0,
                           ptr<double>(c->prices), 1, ptr<int>(c->status));
        if ((rc = from_device(c, p->memspace, prices_out, ptr<double>(c->prices), n))) return rc;
        hstatus.resize(n);
        HIP_TRY(c, hipMemcpyAsync(hstatus.data(), c->status.p, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipGetLastError());
    if ((rc = finish_timing(c, d, pl))) return rc;
    for (int k = 0; k < (int)hstatus.size(); k++)
        if (hstatus[k]) return fail(c, HADI_ERR_NOT_ON_GRID, "S_0 = %.17g is not a node of instance %d's s-grid", S_0, k);
    return HADI_OK;
}

// compute_jacobian*: the reference runs 6 solves one after the other inside each team
// (jacobian_computation.cpp:232-363); here they are 6n independent instances of ONE batched sweep:
// group 0 = base, 1..4 = kappa, eta, sigma, rho + eps, 5 = v-grid rebuilt for V_0 + eps.
int jacobian_common(Ctx *c, const hadi_problem *p, double S_0, double V_0, double eps, double *J, double *base_prices) {
    int rc = check_problem(c, p, false, false);
    if (rc) return rc;
    DeviceGuard guard(c->device);
    pin_rewind(c);
    if (!p->U_0) return fail(c, HADI_ERR_INVALID, "U_0 (initial condition) is required for the Jacobian");
    if (!J || !base_prices) return fail(c, HADI_ERR_INVALID, "J / base_prices missing");
    const int n0 = p->n_instances, m1 = p->m1, m2 = p->m2, G = 6;
    if (!c->device_vgrid && p->V_0_i)
        return fail(c, HADI_ERR_INVALID, "V_0_i needs the device v-grid rebuild (hadi_set_tuning \"device_vgrid\", 1)");
    const int n = n0 * G;
    const size_t m = (size_t)(m1 + 1) * (m2 + 1);
    SweepDesc d;
    fill_common(p, d);
    d.n = n; d.n_src = n0;
    fill_par(p, d, G);
    for (int k = 0; k < n0; k++) {
        d.par8[((size_t)1 * n0 + k) * 8 + 2] += eps;  // kappa
        d.par8[((size_t)2 * n0 + k) * 8 + 3] += eps;  // eta
        d.par8[((size_t)3 * n0 + k) * 8 + 1] += eps;  // sigma
        d.par8[((size_t)4 * n0 + k) * 8 + 0] += eps;  // rho
    }
    // s-grids: replicate the caller's rows into the 6 groups
    const double *src_s, *src_ds;
    if ((rc = to_device(c, p->memspace, p->vec_s, (size_t)n0 * (m1 + 1), c->natU, &src_s))) return rc;
    if ((rc = to_device(c, p->memspace, p->delta_s, (size_t)n0 * m1, c->natOut, &src_ds))) return rc;
    std::vector<int> sel_a(n), sel_b(n);
    std::vector<double> v0i(n);
    for (int g = 0; g < G; g++)
        for (int k = 0; k < n0; k++) {
            sel_a[g * n0 + k] = k;
            sel_b[g * n0 + k] = (g == 5) ? 1 : 0;
            const double v0k = p->V_0_i ? p->V_0_i[k] : V_0;
            v0i[g * n0 + k] = (g == 5) ? v0k + eps : v0k;
        }
    if ((rc = ensure(c, c->sel_a, n * sizeof(int))) || (rc = ensure(c, c->sel_b, n * sizeof(int))) ||
        (rc = ensure(c, c->v0_i, n * 8)) || (rc = ensure(c, c->src_v, 2 * (m2 + 1) * 8)) ||
        (rc = ensure(c, c->src_dv, 2 * m2 * 8)) || (rc = ensure(c, c->g_s, (size_t)n * (m1 + 1) * 8)) ||
        (rc = ensure(c, c->g_ds, (size_t)n * m1 * 8)) || (rc = ensure(c, c->g_v, (size_t)n * (m2 + 1) * 8)) ||
        (rc = ensure(c, c->g_dv, (size_t)n * m2 * 8)))
        return rc;
    hipStream_t s = c->stream;
    if ((rc = stage_to_device(c, c->sel_a.p, sel_a.data(), n * sizeof(int)))) return rc;
    hipLaunchKernelGGL(hadi_bcast_rows_kernel, dim3(grid1d((size_t)n * (m1 + 1))), dim3(256), 0, s, m1 + 1, n, src_s,
                       ptr<int>(c->sel_a), ptr<double>(c->g_s));
    hipLaunchKernelGGL(hadi_bcast_rows_kernel, dim3(grid1d((size_t)n * m1)), dim3(256), 0, s, m1, n, src_ds,
                       ptr<int>(c->sel_a), ptr<double>(c->g_ds));
    std::vector<double> hv(2 * (m2 + 1)), hdv(2 * m2);
    if (c->device_vgrid) {
        // every instance rebuilds its own v-grid on the device: V_0 for the groups 0..4, V_0 + eps for group 5
        // (jacobian_computation.cpp:253,339-341)
        if ((rc = rebuild_v_device(c, n, m2, v0i))) return rc;
    } else {
        build_v(m2, V_0, 5.0, 5.0 / 500, hv.data(), hdv.data());
        build_v(m2, V_0 + eps, 5.0, 5.0 / 500, hv.data() + m2 + 1, hdv.data() + m2);
        if ((rc = stage_to_device(c, c->sel_b.p, sel_b.data(), n * sizeof(int))) || (rc = stage_to_device(c, c->v0_i.p, v0i.data(), (size_t)n * 8)) ||
            (rc = stage_to_device(c, c->src_v.p, hv.data(), (size_t)2 * (m2 + 1) * 8)) || (rc = stage_to_device(c, c->src_dv.p, hdv.data(), (size_t)2 * m2 * 8)))
            return rc;
        hipLaunchKernelGGL(hadi_bcast_rows_kernel, dim3(grid1d((size_t)n * (m2 + 1))), dim3(256), 0, s, m2 + 1, n,
                           ptr<double>(c->src_v), ptr<int>(c->sel_b), ptr<double>(c->g_v));
        hipLaunchKernelGGL(hadi_bcast_rows_kernel, dim3(grid1d((size_t)n * m2)), dim3(256), 0, s, m2, n,
                           ptr<double>(c->src_dv), ptr<int>(c->sel_b), ptr<double>(c->g_dv));
    }
    // (no synchronisation here: the host vectors above went through the pinned arena, and the device-side staging buffers natU /
    // natOut are only reused by later operations of the same stream)
    d.d_vec_s = ptr<double>(c->g_s); d.d_delta_s = ptr<double>(c->g_ds);
    d.d_vec_v = ptr<double>(c->g_v); d.d_delta_v = ptr<double>(c->g_dv);
    // every solve starts from U_0 (jacobian_computation.cpp:307-309); payoff for American = U_0 too
    if ((rc = to_device(c, p->memspace, p->U_0, n0 * m, c->natU0, &d.d_natU))) return rc;
    d.d_natU0 = d.d_natU;

    HadiPlan pl;
    if ((rc = run_sweep(c, d, pl))) return rc;
    if ((rc = ensure(c, c->prices, n * 8)) || (rc = ensure(c, c->status, n * sizeof(int)))) return rc;
    hipLaunchKernelGGL(hadi_pick_kernel, dim3((n + 63) / 64), dim3(64), 0, s, pl.L, n, d.d_vec_s, d.d_vec_v,
                       ptr<double>(c->U), S_0, ptr<double>(c->v0_i), V_0, ptr<double>(c->prices), 1, ptr<int>(c->status));
    // J(k, param) = (pert - base) / eps on the device (jacobian_computation.cpp:329,360): with HADI_MEM_DEVICE the rows
    // never leave HBM (hadi_lm_partials_device reduces them there); only the n status words come back
    double *dJ = J, *db = base_prices;
    if (p->memspace == HADI_MEM_HOST) {
        if ((rc = ensure(c, c->natOut, (size_t)n0 * 6 * 8))) return rc;
        dJ = ptr<double>(c->natOut);
        db = dJ + (size_t)n0 * 5;
    }
    hipLaunchKernelGGL(hadi_jacobian_rows_kernel, dim3((n0 + 255) / 256), dim3(256), 0, s, n0, ptr<double>(c->prices), eps, dJ, db);
    if (p->memspace == HADI_MEM_HOST) {
        HIP_TRY(c, hipMemcpyAsync(J, dJ, (size_t)n0 * 5 * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(base_prices, db, (size_t)n0 * 8, hipMemcpyDeviceToHost, s));
    }
    std::vector<int> hs_fallback;
    int *hs = static_cast<int *>(pin_alloc(c, (size_t)n * sizeof(int)));  // (pinned: the copy does not block the host)
    if (!hs) { hs_fallback.resize(n); hs = hs_fallback.data(); }
    HIP_TRY(c, hipMemcpyAsync(hs, c->status.p, n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipGetLastError());
    if ((rc = finish_timing(c, d, pl))) return rc;
    for (int k = 0; k < n0; k++)
        if (hs[k]) return fail(c, HADI_ERR_NOT_ON_GRID, "S_0 = %.17g is not a node of instance %d's s-grid", S_0, k);
    return HADI_OK;
}

int with_variant(hadi_ctx *ctx, const hadi_problem *p, int variant, hadi_problem *tmp) {
    if (!ctx) return HADI_ERR_INVALID;
    if (!p) return fail(reinterpret_cast<Ctx *>(ctx), HADI_ERR_INVALID, "problem is NULL");
    *tmp = *p;
    tmp->variant = variant;
    return HADI_OK;
}

// Frees everything a (possibly half-built) handle owns.  The caller has made the handle's device current.
void release_handle(Ctx *c) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf *bufs[] = {&c->U, &c->Y, &c->LAM, &c->U0, &c->UT, &c->scoef, &c->b2row, &c->rowc, &c->a2i, &c->pb,
                      &c->rinv, &c->rwork, &c->ipar, &c->par8, &c->g_s, &c->g_v, &c->g_ds, &c->g_dv, &c->src_v,
                      &c->src_dv, &c->sel_a, &c->sel_b, &c->v0_i, &c->natU, &c->natU0, &c->natOut, &c->prices,
                      &c->status, &c->div_flag, &c->div_amt, &c->div_pct, &c->V, &c->R1, &c->C2, &c->pay_mis, &c->Uf, &c->Yf,
                      &c->order, &c->lm31, &c->team};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto &g : c->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    for (auto e : c->kev) (void)hipEventDestroy(e);
    for (auto e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->wait_ev) (void)hipEventDestroy(c->wait_ev);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->join_ev) (void)hipEventDestroy(c->join_ev);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->err_host) (void)hipHostFree(c->err_host);
    if (c->pin) (void)hipHostFree(c->pin);
    delete c;
}

}  // namespace

// =====================================================================================================
extern "C" {

int hadi_version(void) { return HADI_VERSION_MAJOR * 100 + HADI_VERSION_MINOR; }

#if defined(HADI_TEAM_STAMPS)
// diagnostic build only (tools/team_stamps.py): the 16 phase stamps of the last instance-resident launch
int hadi_debug_team_stamps(hadi_ctx *ctx, unsigned long long *out16) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c || !c->team.p) return 1;
    return hipMemcpy(out16, ptr<int>(c->team) + 384, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess;
}
#endif

#if defined(HADI_STAMPS)
// diagnostic build only (tools/stamps.py)
int hadi_debug_stamps(unsigned long long *out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_hadi_stamps), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_hadi_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

const char *hadi_status_string(int s) {
    switch (s) {
        case HADI_OK: return "ok";
        case HADI_ERR_INVALID: return "invalid argument";
        case HADI_ERR_UNSUPPORTED: return "unsupported grid shape";
        case HADI_ERR_HIP: return "HIP runtime error";
        case HADI_ERR_NOT_ON_GRID: return "S_0 is not a grid node";
        case HADI_ERR_NO_DEVICE: return "no usable gfx950 GPU (libhadi has no CPU path)";
        case HADI_ERR_ALLOC: return "device allocation failed";
        case HADI_ERR_INTERNAL: return "device-side failure reported by a kernel (results invalid)";
        default: return "unknown";
    }
}

int hadi_create(hadi_ctx **out, int device_id) {
    if (!out) return HADI_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return HADI_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return HADI_ERR_INVALID;
    DeviceGuard guard(device_id);  // the caller's current device is left as it was found
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return HADI_ERR_HIP;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return HADI_ERR_NO_DEVICE;  // code object is gfx950-only
    Ctx *c = new Ctx;
    c->device = device_id;
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->name = prop.name;
    c->arch = prop.gcnArchName;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && raise_all_lds_limits() == hipSuccess;
    for (auto &e : c->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->wait_ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&c->err_host), 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    if (ok) {
        c->pin_want = (size_t)1 << 20;  // 1 MB to start with (3000 instances' parameter rows are 192 KB)
        pin_rewind(c);
        *c->err_host = 0;
        ok = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->err_dev), c->err_host, 0) == hipSuccess;
    }
    if (!ok) {  // one way out: whatever was created is released
        release_handle(c);
        return HADI_ERR_HIP;
    }
    *out = reinterpret_cast<hadi_ctx *>(c);
    return HADI_OK;
}

int hadi_destroy(hadi_ctx *ctx) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return HADI_OK;
    DeviceGuard guard(c->device);
    release_handle(c);
    return HADI_OK;
}

const char *hadi_last_error(const hadi_ctx *ctx) {
    const Ctx *c = reinterpret_cast<const Ctx *>(ctx);
    return c ? c->err.c_str() : "null handle";
}

int hadi_set_profiling(hadi_ctx *ctx, int enabled) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return HADI_ERR_INVALID;
    c->profiling = enabled ? 1 : 0;
    return HADI_OK;
}

int hadi_set_tuning(hadi_ctx *ctx, const char *key, int value) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c || !key) return HADI_ERR_INVALID;
    if (!std::strcmp(key, "graph")) c->use_graph = value ? 1 : 0;
    else if (!std::strcmp(key, "small_grid")) c->use_small = value ? 1 : 0;
    else if (!std::strcmp(key, "small_seq")) c->small_seq = value < 0 ? -1 : (value ? 1 : 0);
    else if (!std::strcmp(key, "american_p")) c->use_amp = value ? 1 : 0;
    else if (!std::strcmp(key, "device_vgrid")) c->device_vgrid = value ? 1 : 0;
    else if (!std::strcmp(key, "sub_batch")) c->sub_batch = value ? 1 : 0;
    else if (!std::strcmp(key, "small_pairs")) c->small_pairs = value < 0 ? -1 : (value ? 1 : 0);
    else if (!std::strcmp(key, "streams")) c->streams = value == 2 ? 2 : (value == 1 ? 1 : 0);
    else if (!std::strcmp(key, "col_prefetch")) c->col_prefetch = value ? 1 : 0;
    else if (!std::strcmp(key, "cs_strips")) c->cs_strips = (value >= 0 && value <= 3) ? value : 1;
    else if (!std::strcmp(key, "graph_max_melems")) c->graph_max_melems = value > 0 ? value : 0;
    else if (!std::strcmp(key, "tile_interleave")) c->tile_il = value ? 1 : 0;
    else if (!std::strcmp(key, "strip")) c->tune.strip = value < 0 ? -1 : (value ? 1 : 0);
    else if (!std::strcmp(key, "debug_fault")) c->debug_fault = value;
    else if (!std::strcmp(key, "team_launch")) { c->team_launch = value < 0 ? -1 : (value ? 1 : 0); c->team_failed = 0; }
    else if (!std::strcmp(key, "row_tile")) c->tune.row_tile = value > 0 ? value : 0;
    else if (!std::strcmp(key, "strip_blocks")) c->tune.strip_blocks = value > 0 ? value : 0;
    else if (!std::strcmp(key, "pair_strips")) c->tune.pair_strips = value < 0 ? -1 : (value ? 1 : 0);
    else if (!std::strcmp(key, "col_groups")) c->tune.col_groups = value > 0 ? value : 0;
    else if (!std::strncmp(key, "model_", 6)) {  // constants of the plan's cost model (hadi_plan.h)
        struct { const char *k; int HadiTuning::*f; } const tab[] = {
            {"model_strip_row_ns", &HadiTuning::strip_row_ns}, {"model_ring_row_ps", &HadiTuning::ring_row_ps}, {"model_ring_fixed_ns", &HadiTuning::ring_fixed_ns},
            {"model_pstrip_row_ns", &HadiTuning::pstrip_row_ns}, {"model_pring_row_ps", &HadiTuning::pring_row_ps}, {"model_pring_fixed_ns", &HadiTuning::pring_fixed_ns}};
        bool hit = false;
        for (auto &e : tab)
            if (!std::strcmp(key, e.k)) { if (value < 1) return fail(c, HADI_ERR_INVALID, "%s must be positive", key); c->tune.*(e.f) = value; hit = true; }
        if (!hit) return fail(c, HADI_ERR_INVALID, "unknown tuning key '%s'", key);
    }
    else if (!std::strcmp(key, "small_waves")) {
        if (value != 0 && value != 4 && value != 8) return fail(c, HADI_ERR_INVALID, "small_waves must be 0, 4 or 8");
        c->tune.small_waves = value;
    } else return fail(c, HADI_ERR_INVALID, "unknown tuning key '%s'", key);
    return HADI_OK;
}

int hadi_get_tuning(const hadi_ctx *ctx, const char *key, int *value) {
    const Ctx *c = reinterpret_cast<const Ctx *>(ctx);
    if (!c || !key || !value) return HADI_ERR_INVALID;
    if (!std::strcmp(key, "graph")) *value = c->use_graph;
    else if (!std::strcmp(key, "small_grid")) *value = c->use_small;
    else if (!std::strcmp(key, "small_seq")) *value = c->small_seq;
    else if (!std::strcmp(key, "american_p")) *value = c->use_amp;
    else if (!std::strcmp(key, "device_vgrid")) *value = c->device_vgrid;
    else if (!std::strcmp(key, "sub_batch")) *value = c->sub_batch;
    else if (!std::strcmp(key, "small_pairs")) *value = c->small_pairs;
    else if (!std::strcmp(key, "streams")) *value = c->streams;
    else if (!std::strcmp(key, "col_prefetch")) *value = c->col_prefetch;
    else if (!std::strcmp(key, "cs_strips")) *value = c->cs_strips;
    else if (!std::strcmp(key, "graph_max_melems")) *value = c->graph_max_melems;
    else if (!std::strcmp(key, "tile_interleave")) *value = c->tile_il;
    else if (!std::strcmp(key, "strip")) *value = c->tune.strip;
    else if (!std::strcmp(key, "debug_fault")) *value = c->debug_fault;
    else if (!std::strcmp(key, "team_launch")) *value = c->team_failed ? -2 : c->team_launch;
    else if (!std::strcmp(key, "row_tile")) *value = c->tune.row_tile;
    else if (!std::strcmp(key, "strip_blocks")) *value = c->tune.strip_blocks;
    else if (!std::strcmp(key, "pair_strips")) *value = c->tune.pair_strips;
    else if (!std::strcmp(key, "col_groups")) *value = c->tune.col_groups;
    else if (!std::strcmp(key, "model_strip_row_ns")) *value = c->tune.strip_row_ns;
    else if (!std::strcmp(key, "model_ring_row_ps")) *value = c->tune.ring_row_ps;
    else if (!std::strcmp(key, "model_ring_fixed_ns")) *value = c->tune.ring_fixed_ns;
    else if (!std::strcmp(key, "model_pstrip_row_ns")) *value = c->tune.pstrip_row_ns;
    else if (!std::strcmp(key, "model_pring_row_ps")) *value = c->tune.pring_row_ps;
    else if (!std::strcmp(key, "model_pring_fixed_ns")) *value = c->tune.pring_fixed_ns;
    else if (!std::strcmp(key, "small_waves")) *value = c->tune.small_waves;
    else return HADI_ERR_INVALID;
    return HADI_OK;
}

int hadi_wait_stream(hadi_ctx *ctx, void *producer_stream) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return HADI_ERR_INVALID;
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipEventRecord(c->wait_ev, static_cast<hipStream_t>(producer_stream)));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->wait_ev, 0));
    return HADI_OK;
}

int hadi_get_timing(const hadi_ctx *ctx, hadi_timing *out) {
    const Ctx *c = reinterpret_cast<const Ctx *>(ctx);
    if (!c || !out) return HADI_ERR_INVALID;
    *out = c->timing;
    return HADI_OK;
}

int hadi_device_info(const hadi_ctx *ctx, char *name, int name_len, int *compute_units, char *arch, int arch_len) {
    const Ctx *c = reinterpret_cast<const Ctx *>(ctx);
    if (!c) return HADI_ERR_INVALID;
    if (name && name_len > 0) std::snprintf(name, name_len, "%s", c->name.c_str());
    if (arch && arch_len > 0) std::snprintf(arch, arch_len, "%s", c->arch.c_str());
    if (compute_units) *compute_units = c->cu_count;
    return HADI_OK;
}

int hadi_describe_last_sweep(const hadi_ctx *ctx, char *buf, int len) {
    const Ctx *c = reinterpret_cast<const Ctx *>(ctx);
    if (!c || !buf || len <= 0) return HADI_ERR_INVALID;
    std::snprintf(buf, len, "%s", c->last_path.c_str());
    return HADI_OK;
}

void *hadi_stream(hadi_ctx *ctx) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    return c ? (void *)c->stream : nullptr;
}

// ---- grids ------------------------------------------------------------------------------------------
int hadi_make_grid(int m1, double S, double S_0, double K, double cc, int m2, double V, double V_0, double d,
                   double *vec_s, double *vec_v, double *delta_s, double *delta_v) {
    if (m1 < 1 || m2 < 1 || !vec_s || !vec_v || !delta_s || !delta_v) return HADI_ERR_INVALID;
    const double lo = std::asinh(-K / cc);
    const double Delta_xi = (1.0 / m1) * (std::asinh((S - K) / cc) - lo);
    for (int i = 0; i <= m1; i++) vec_s[i] = K + cc * std::sinh(lo + i * Delta_xi);
    sorted_insert_drop_largest(vec_s, m1 + 1, S_0);
    for (int i = 0; i < m1; i++) delta_s[i] = vec_s[i + 1] - vec_s[i];
    build_v(m2, V_0, V, d, vec_v, delta_v);
    return HADI_OK;
}

int hadi_rebuild_variance(int m2, double V_0_new, double V, double d, double *vec_v, double *delta_v) {
    if (m2 < 1 || !vec_v || !delta_v) return HADI_ERR_INVALID;
    build_v(m2, V_0_new, V, d, vec_v, delta_v);
    return HADI_OK;
}

int hadi_find_s_index(int m1, const double *vec_s, double S_0) {
    for (int i = 0; i <= m1; i++)
        if (std::fabs(vec_s[i] - S_0) < 1e-10) return i;
    return -1;
}

int hadi_find_v_index(int m2, const double *vec_v, double V_0) {
    for (int i = 0; i <= m2; i++)
        if (std::fabs(vec_v[i] - V_0) < 1e-10) return i;
    return 0;
}

// ---- hot path ------------------------------------------------------------------------------------------
int hadi_DO_timestepping(hadi_ctx *ctx, const hadi_problem *p) {
    return solve_common(reinterpret_cast<Ctx *>(ctx), p, false, false, 0.0, 0.0, nullptr);
}

int hadi_parallel_DO_solve(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0, double *base_prices) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (c && !base_prices) return fail(c, HADI_ERR_INVALID, "base_prices missing");
    return solve_common(c, p, false, true, S_0, V_0, base_prices);
}

int hadi_compute_base_prices(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0, double *base_prices) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (c && !base_prices) return fail(c, HADI_ERR_INVALID, "base_prices missing");
    return solve_common(c, p, true, true, S_0, V_0, base_prices);
}

int hadi_debug_row_pass(hadi_ctx *ctx, const hadi_problem *p, int step, double *Y1rhs) {
    return solve_common(reinterpret_cast<Ctx *>(ctx), p, false, false, 0.0, 0.0, nullptr, 1, step, Y1rhs);
}

int hadi_debug_col_solve(hadi_ctx *ctx, const hadi_problem *p, double *X) {
    return solve_common(reinterpret_cast<Ctx *>(ctx), p, false, false, 0.0, 0.0, nullptr, 2, 1, X);
}

int hadi_debug_rcp(hadi_ctx *ctx, int n, const double *x, double *out) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return HADI_ERR_INVALID;
    if (n < 1 || !x || !out) return fail(c, HADI_ERR_INVALID, "bad arguments");
    DeviceGuard guard(c->device);
    int rc;
    if ((rc = ensure(c, c->natU, (size_t)n * 8)) || (rc = ensure(c, c->natOut, (size_t)n * 8))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->natU.p, x, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(hadi_rcp_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, c->stream, n, ptr<double>(c->natU), ptr<double>(c->natOut));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->natOut.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HADI_OK;
}

#define HADI_VARIANT_WRAPPERS(suffix, variant)                                                                   \
    int hadi_compute_base_prices_##suffix(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0,          \
                                          double *base_prices) {                                                 \
        hadi_problem t;                                                                                          \
        int rc = with_variant(ctx, p, variant, &t);                                                              \
        return rc ? rc : hadi_compute_base_prices(ctx, &t, S_0, V_0, base_prices);                               \
    }                                                                                                            \
    int hadi_compute_jacobian_##suffix(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0, double eps, \
                                       double *J, double *base_prices) {                                         \
        hadi_problem t;                                                                                          \
        int rc = with_variant(ctx, p, variant, &t);                                                              \
        return rc ? rc : hadi_compute_jacobian(ctx, &t, S_0, V_0, eps, J, base_prices);                          \
    }

int hadi_compute_jacobian(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0, double eps, double *J,
                          double *base_prices) {
    return jacobian_common(reinterpret_cast<Ctx *>(ctx), p, S_0, V_0, eps, J, base_prices);
}

HADI_VARIANT_WRAPPERS(american, HADI_AM)
HADI_VARIANT_WRAPPERS(dividends, HADI_DIV)
HADI_VARIANT_WRAPPERS(american_dividends, HADI_AM_DIV)

// ---- Levenberg-Marquardt normal equations (jacobian_computation.cpp:20-195) ---------------------------
int hadi_lm_partials(int n, const double *J, const double *r, double *out) {
    if (n < 0 || !out || (n > 0 && (!J || !r))) return HADI_ERR_INVALID;
    for (int k = 0; k < 31; k++) out[k] = 0.0;
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) {
            double s = 0.0;
            for (int k = 0; k < n; k++) s += J[(size_t)k * 5 + i] * J[(size_t)k * 5 + j];
            out[i * 5 + j] = s;
        }
    for (int i = 0; i < 5; i++) {
        double s = 0.0;
        for (int k = 0; k < n; k++) s += J[(size_t)k * 5 + i] * r[k];
        out[25 + i] = s;
    }
    double s2 = 0.0;
    for (int k = 0; k < n; k++) s2 += r[k] * r[k];
    out[30] = s2;
    return HADI_OK;
}

int hadi_lm_partials_device(hadi_ctx *ctx, int n, const double *J, const double *model_prices, const double *market_prices,
                            double *partial31) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return HADI_ERR_INVALID;
    if (n < 0 || !partial31 || (n > 0 && (!J || !model_prices || !market_prices))) return fail(c, HADI_ERR_INVALID, "bad arguments");
    DeviceGuard guard(c->device);
    int rc = ensure(c, c->lm31, 31 * sizeof(double));
    if (rc) return rc;
    hipLaunchKernelGGL(hadi_lm_partials_kernel, dim3(1), dim3(256), (size_t)21 * 256 * sizeof(double), c->stream, n, J, model_prices,
                       market_prices, ptr<double>(c->lm31));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(partial31, c->lm31.p, 31 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HADI_OK;
}

int hadi_lm_solve(const double *partial31, double lambda, double *delta5) {
    if (!partial31 || !delta5) return HADI_ERR_INVALID;
    constexpr int NP = 5;
    double A[NP * NP], b[NP];
    for (int i = 0; i < NP * NP; i++) A[i] = partial31[i];
    for (int i = 0; i < NP; i++) {
        A[i * NP + i] *= (1.0 + lambda);  // jacobian_computation.cpp:136-138
        b[i] = partial31[25 + i];
    }
    for (int k = 0; k < NP; k++) {  // partial-pivot elimination, jacobian_computation.cpp:44-83
        int piv = k;
        double best = std::fabs(A[k * NP + k]);
        for (int r = k + 1; r < NP; r++)
            if (std::fabs(A[r * NP + k]) > best) { best = std::fabs(A[r * NP + k]); piv = r; }
        if (piv != k) {
            for (int col = 0; col < NP; col++) std::swap(A[k * NP + col], A[piv * NP + col]);
            std::swap(b[k], b[piv]);
        }
        const double pv = A[k * NP + k];
        for (int col = k + 1; col < NP; col++) A[k * NP + col] /= pv;
        b[k] /= pv;
        A[k * NP + k] = 1.0;
        for (int i = k + 1; i < NP; i++) {
            const double f = A[i * NP + k];
            for (int col = k + 1; col < NP; col++) A[i * NP + col] -= f * A[k * NP + col];
            b[i] -= f * b[k];
            A[i * NP + k] = 0.0;
        }
    }
    for (int k = NP - 1; k >= 0; k--) {
        double v = b[k];
        for (int col = k + 1; col < NP; col++) v -= A[k * NP + col] * b[col];
        b[k] = v;
    }
    for (int i = 0; i < NP; i++) delta5[i] = b[i];
    return HADI_OK;
}

int hadi_compute_parameter_update(int n, const double *J, const double *r, double lambda, double *delta5) {
    double part[31];
    int rc = hadi_lm_partials(n, J, r, part);
    if (rc) return rc;
    return hadi_lm_solve(part, lambda, delta5);
}

}  // extern "C"
