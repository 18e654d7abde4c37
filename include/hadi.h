/*
 * hadi.h -- C ABI of libhadi: the MI355X-native batched Heston Douglas-ADI time stepper.
 *
 * Drop-in boundary for the hot path of BCW-dot/PDE-based-Heston-Solver-GPU-accelerated.
 * The reference has no FFI: its hot path is reached through the host-callable batched
 * launchers listed below (Kokkos templates).  Each entry point here takes, as flat arrays,
 * exactly the data those launchers take as Kokkos Views / struct arrays
 * (GridViews: grid_pod.hpp:8-15; DO_Workspace.U: DO_solver_workspace.hpp:7; U_0; dividend
 * views; model / market / numerics scalars), so a maintainer can bind it from the reference
 * host code with `view.data()` pointers -- see INTEGRATION.md.
 *
 *   hadi_DO_timestepping            <- device_DO_timestepping{,_american,_dividend,_american_dividend}
 *                                      (src/device_solver.hpp:194-266, 274-374, 382-641, 650-942)
 *                                      + Device_BoundaryConditions::initialize (hes_boundary_kernels.hpp:41-75)
 *                                      + Device_A{0,1,2}::build_matrix (hes_a0_kernels.hpp:30,
 *                                        hes_a1_kernels.hpp:51, hes_a2_shuffled_kernels.hpp:103)
 *                                      with scheme = HADI_SCHEME_CRAIG_SNEYD: CS_scheme_shuffled (src/solver.hpp:781-907)
 *   hadi_parallel_DO_solve          <- parallel_DO_solve (src/device_solver.hpp:52-185)
 *   hadi_compute_base_prices[_*]    <- compute_base_prices{,_american,_dividends,_american_dividends}
 *                                      (src/jacobian_computation.cpp:368, 629, 922, 1232)
 *   hadi_compute_jacobian[_*]       <- compute_jacobian{,_american,_dividends,_american_dividends}
 *                                      (src/jacobian_computation.cpp:204, 457, 726, 1031)
 *   hadi_compute_parameter_update   <- compute_parameter_update_on_device + solve_5x5_device
 *                                      (src/jacobian_computation.cpp:20-195)
 *   hadi_make_grid / hadi_rebuild_variance <- Grid::Grid (src/grid.cpp:16-61),
 *                                      GridViews::rebuild_variance_views (src/grid_pod.hpp:25-73)
 *
 * Conventions
 *   - All arithmetic is IEEE fp64.  Logical layout of every [n][m] field is the reference's:
 *     idx = i + j*(m1+1), i = s-index (fastest), j = v-index, m = (m1+1)(m2+1)
 *     (src/device_solver.cpp:713).  The library keeps its own internal HBM layout.
 *   - Array arguments live in the memory space named by hadi_problem.memspace
 *     (HADI_MEM_DEVICE: device pointers, data already resident in HBM; HADI_MEM_HOST: host
 *     pointers, the library stages them).  Scalars, per-instance parameter overrides and the
 *     dividend schedule are always host memory.  Outputs (base_prices, J) follow memspace.
 *   - Every call is synchronous on the handle's stream (the reference fences before returning:
 *     device_solver.hpp:184, jacobian_computation.cpp:363,447).  A handle is not thread-safe.
 *   - Return value: HADI_OK or an error code; hadi_last_error() gives the message.  The
 *     reference has `void` returns and silently reads index -1 when S_0 is not a grid node
 *     (jacobian_computation.cpp:275-288); here that is HADI_ERR_NOT_ON_GRID.
 *   - There is no CPU fallback: without a usable gfx950 device hadi_create fails.
 */
#ifndef HADI_H
#define HADI_H

#ifdef __cplusplus
extern "C" {
#endif

#define HADI_VERSION_MAJOR 0
#define HADI_VERSION_MINOR 2

typedef struct hadi_ctx hadi_ctx;

enum hadi_status {
    HADI_OK = 0,
    HADI_ERR_INVALID = 1,      /* bad argument (NULL, sizes, variant) */
    HADI_ERR_UNSUPPORTED = 2,  /* grid shape outside what the kernels cover */
    HADI_ERR_HIP = 3,          /* a HIP runtime call failed */
    HADI_ERR_NOT_ON_GRID = 4,  /* S_0 is not a node of some instance's s-grid */
    HADI_ERR_NO_DEVICE = 5,    /* no usable GPU: the product has no CPU path */
    HADI_ERR_ALLOC = 6,
    HADI_ERR_INTERNAL = 7      /* a kernel reported a failure through the handle's device error word (e.g. the bounded
                                  rendezvous of a two-wavefront row ran out of polls): the outputs of the call are invalid */
};

enum hadi_variant { HADI_EU = 0, HADI_AM = 1, HADI_DIV = 2, HADI_AM_DIV = 3 };
enum hadi_memspace { HADI_MEM_HOST = 0, HADI_MEM_DEVICE = 1 };
/* Splitting scheme.  The reference's device path is Douglas only (src/device_solver.hpp:194-266); Craig-Sneyd
 * exists in its host family (CS_scheme_shuffled, src/solver.hpp:781-907, European) and is offered here on the
 * device for the European variant. */
enum hadi_scheme { HADI_SCHEME_DOUGLAS = 0, HADI_SCHEME_CRAIG_SNEYD = 1 };
/* Precision of the state arrays BETWEEN the two directional passes.  FP64 is what the reference computes in.  FP32
 * ("mixed-precision fp32 ADI sweep with fp64 tridiag pivots", BASELINE.json config 5): U and the A2 right-hand side are
 * stored as fp32 in HBM (half the traffic: 16 B per point-step), every operator, pivot and line solve is still
 * evaluated in fp64 in registers.  European Douglas sweeps (HADI_EU, HADI_DIV) only; the caller's arrays stay fp64.
 * Not a reference feature: parity is against the oracle run with the same two roundings per step (tests).  Each rounding
 * perturbs the state by 2^-24 relative and the perturbations add up like a random walk over the steps: against the
 * fp64 sweep the price moves by 1e-7 .. 1.5e-5 (erratic in N: one realisation of the walk per run; 1.5e-5 at
 * 1024x512x2000) and the field by up to 1.4e-5 of max|U| at N = 2000 (increments below half an fp32 ulp are absorbed by
 * the rounding) -- a throughput mode for users who accept that; it does NOT guarantee a 1e-6 price tolerance. */
enum hadi_state_precision { HADI_STATE_FP64 = 0, HADI_STATE_FP32 = 1 };
/* Boundary data of the option type.  HADI_CALL is the reference's only boundary class (call-specific,
 * src/BoundaryConditions.hpp:7-12, hes_boundary_kernels.hpp:41-75).  HADI_PUT is NOT a reference feature (README.md:26
 * claims puts, no code exists): the same operators with the put's boundary data --
 *   s = s_max:  du/ds = 0             -> b1 == 0                  (call: du/ds = e^{-r_f t} -> b1 = (r_d - r_f) s_max E)
 *   v = v_max:  u = K e^{-r_d t}      -> b2 = -1/2 r_d K on the last v-row, time factor e^{-r_d dt n}
 *                                                                  (call: u = s e^{-r_f t} -> b2 = -1/2 r_d s_i E)
 *   s = 0:      u = K e^{-r_d t}      -> the i = 0 row of A1 carries the reaction term -1/2 r_d like every other row
 *                                        (the call leaves that row empty, hes_a1_kernels.hpp:56-61: its value stays 0)
 *   v = 0 and the two top v-rows: the PDE rows / empty rows of A2 exactly as for the call.
 * Needs hadi_problem.strike_i.  Validated by put-call parity against the call path and the semi-analytic Heston price. */
enum hadi_option_type { HADI_CALL = 0, HADI_PUT = 1 };

/* One batch of independent option instances = one league of teams in the reference
 * (TeamPolicy(nInstances, AUTO), device_solver.hpp:83-88). */
typedef struct hadi_problem {
    int n_instances;
    int m1, m2;            /* grid intervals: (m1+1) s-nodes, (m2+1) v-nodes */
    int variant;           /* enum hadi_variant */
    int memspace;          /* enum hadi_memspace, applies to the array fields below */

    /* time discretisation (shared); N_i / delta_t_i override per instance (multi-maturity,
     * heston_calibration.cpp:2165-2171), host arrays of length n_instances or NULL */
    int N;
    double delta_t;
    double theta;
    const int *N_i;
    const double *delta_t_i;

    /* market + Heston parameters (shared across the batch in the reference,
     * jacobian_computation.cpp:206-208); *_i override per instance (host arrays) or NULL */
    double r_d, r_f;
    double rho, sigma, kappa, eta;
    const double *rho_i, *sigma_i, *kappa_i, *eta_i;

    /* GridViews, one row per instance (grid_pod.hpp:9-15) */
    const double *vec_s;   /* [n][m1+1] */
    const double *vec_v;   /* [n][m2+1] */
    const double *delta_s; /* [n][m1]   */
    const double *delta_v; /* [n][m2]   */

    /* discrete dividends, shared (device_solver.hpp:409-413); host arrays */
    int num_dividends;
    const double *dividend_dates, *dividend_amounts, *dividend_percentages;

    /* DO_Workspace.U: in = initial condition, out = solution at T.  [n][m] */
    double *U;
    /* payoff for the American projection (U_0_i, device_solver.hpp:304); [n][m].
     * NULL = use the initial U. */
    const double *U_0;
    /* optional output: lambda_bar at T (DO_solver_workspace.hpp:22); [n][m] or NULL */
    double *lambda_bar;
    /* enum hadi_scheme; 0 (Douglas) is what every reference launcher runs */
    int scheme;
    /* enum hadi_state_precision; 0 (fp64) is what every reference launcher runs */
    int state_precision;
    /* enum hadi_option_type; 0 (call boundary data) is what every reference launcher runs */
    int option_type;
    /* strikes, host array [n]; required for HADI_PUT (boundary value K e^{-r_d t}), ignored for HADI_CALL */
    const double *strike_i;
    /* hadi_compute_base_prices* / hadi_compute_jacobian* only: per-instance V_0 (host array [n]) overriding the
     * scalar argument -- every instance's v-grid is rebuilt on the device for its own V_0, as every team of the
     * reference does in-kernel (GridViews::rebuild_variance_views, src/grid_pod.hpp:25-73).  NULL = the scalar V_0. */
    const double *V_0_i;
} hadi_problem;

/* Timing of the last sweep on this handle, measured with HIP events on the handle's
 * stream.  Kernel sums are only filled when profiling was enabled with hadi_set_profiling. */
typedef struct hadi_timing {
    double setup_ms;        /* staging + operator/boundary setup + layout pack */
    double sweep_ms;        /* the N-step time loop only */
    double finish_ms;       /* unpack + price pick */
    double pass_a_ms;       /* sum over launches of the row-pass kernel   (profiling only) */
    double pass_b_ms;       /* sum over launches of the column-pass kernel (profiling only) */
    long long pass_a_launches, pass_b_launches;
    long long point_steps;  /* sum over instances of (m1+1)(m2+1)*N_i */
} hadi_timing;

/* ---- handle ----------------------------------------------------------------------------- */
int hadi_create(hadi_ctx **out, int device_id);
int hadi_destroy(hadi_ctx *ctx);
const char *hadi_last_error(const hadi_ctx *ctx);
const char *hadi_status_string(int status);
int hadi_version(void);
int hadi_set_profiling(hadi_ctx *ctx, int enabled);
int hadi_get_timing(const hadi_ctx *ctx, hadi_timing *out);
/* Execution-path switches (results agree to round-off; the library reads NO environment variables):
 *   "small_grid"  LDS-resident one-launch path for grids that fit in LDS (default 1)
 *   "small_seq"   ... European / dividend sweeps of such grids on the one-wavefront-per-instance kernel that solves the
 *                 lines sequentially, one per lane: -1 automatic (default: batches of more instances than CUs), 0 never
 *                 (the block-per-instance kernel, which the American sweeps always use), 1 always
 *   "graph"       hipGraph replay of the time loop for small batches (default 1)
 *   "american_p"  American sweeps keep P = U_bar - dt*lambda_bar in place of U and no lambda_bar array whenever every
 *                 payoff of the batch depends on s only (default 1; 0 = always the explicit (U, lambda_bar) pair)
 *   "strip"       strip row pass: -1 automatic (default), 0 never, 1 whenever the geometry allows it
 *   "row_tile"    shared-ring row pass: v-rows per block tile (0 = automatic)
 *   "strip_blocks" strip row pass: blocks per instance (0 = automatic)
 *   "model_strip_row_ns", "model_ring_row_ps", "model_ring_fixed_ns", "model_pstrip_row_ns", "model_pring_row_ps",
 *   "model_pring_fixed_ns"  constants of the cost model that decides between strips and the shared ring at 8 nodes per lane
 *                 (hadi_plan.h; defaults 2800 / 2330 / 12000 and, for two wavefronts per row, 3250 / 4300 / 15000: measured on
 *                 one MI355X, the boxes of a pool differ by +-3 % on the kernels they model)
 *   "pair_strips" strip row pass at 4 nodes per lane (128 < m1 <= 256): two strips per wavefront on the 8-nodes-per-lane
 *                 arithmetic (hadi_pass_a_pairs): -1 automatic (default: where the launch still has two blocks per CU and the
 *                 strips keep 16 rows, e.g. 512 instances of 256x128), 0 never, 1 wherever the strips run
 *   "col_groups"  column pass: blocks per instance (0 = automatic)
 *   "small_waves" small-grid kernel: wavefronts per instance, 4 or 8 (0 = automatic)
 *   "small_pairs" LDS-resident European / dividend sweeps: two instances per wavefront (hadi_small_seq2_kernel; grids of at most
 *                 32 v-rows): -1 automatic (default: batches of more than 2 and at most 4.5 instances per CU, where it is 14 - 20 %
 *                 faster), 0 never, 1 whenever possible
 *   "sub_batch"   batches of several rounds of one instance per CU on grids whose round exceeds the 256 MB memory-side
 *                 cache run sub-batch by sub-batch through the time loop (default 1; instances are independent)
 *   "streams"     0 (default) automatic, 1 one stream, 2 two streams.  On two streams the two halves of a Douglas batch run their
 *                 time loops side by side (fork / join by events inside the call; per-launch profiling switches it off).
 *                 Same kernels, same results.  It pays where the row pass of the whole batch leaves a partial round of CUs
 *                 idle -- 512x256: 160 instances +7 %, 192: +16 %, 24..32: +14..22 % -- and costs 2..12 % where the launches
 *                 fill whole rounds (64, 128, 256, 512 instances): the automatic choice cuts the batch (or the remainder
 *                 behind its full rounds of one instance per CU) in two exactly when the plan's row pass leaves 4 % or more
 *                 of its CU-rounds idle (hadi_plan_row_idle) -- a function of the grid shape and the batch size alone.
 *                 With 2 the sub-batches of a large batch alternate between the streams -- and so they do in the automatic mode
 *                 when a batch is several sub-batches of strip launches (measured never slower, +3 .. +6 % with a remainder behind
 *                 one full round); DESIGN.md sections 4.2, 7
 *   "col_prefetch" 0 (default) / 1: European sweeps of 9 .. 16 chunks (264 <= m2 <= 527) on hadi_pass_b2 -- part of the next column
 *                 tile prefetched into LDS by LDS-DMA -- instead of hadi_pass_b1; "tile_interleave" 0 (default) / 1: the blocks of
 *                 an instance walk their column tiles interleaved.  Both measured within +-2 % (profiles/r04_colpass_ab.txt);
 *                 same bits as the default path
 *   "cs_strips"   1 (default) / 0: the predictor / corrector row passes of a Craig-Sneyd sweep on the barrier-free strips wherever
 *                 the plan picks strips (0: on the shared ring, the path of rounds 2 - 3; same fields to round-off)
 *   "graph_max_melems" hipGraph replay of the time loop for batches of up to this many Mi state elements (default 8; larger
 *                 batches are not launch-bound: measured equal)
 *   "device_vgrid" v-grids of compute_base_prices / compute_jacobian rebuilt per instance on the device (default 1;
 *                 0 = built once on the host with glibc sinh/asinh and broadcast -- bit-identical to the reference's
 *                 host-side Grid, needs one shared V_0)
 *   "team_launch" instance-resident execution of batches of up to 8 large European instances, with or without discrete dividends
 *                 (128 < m1 <= 512, m2 <= 263):
 *                 the whole time loop in ONE launch, every instance kept in the L2 of one XCD by a team of 32 blocks
 *                 (hadi_team_kernel): -1 automatic (default), 0 never, 1 whenever the shape allows it.  If the team
 *                 protocol fails (it is bounded everywhere) the batch is solved again on the two-launches-per-step path
 *                 and the automatic choice stays away from it for the rest of the handle's life (get: -2)
 *   "debug_fault" TEST HOOK, 0 in production: 1 = the high half of every two-wavefront row (m1 > 512) withholds its
 *                 rendezvous token on v-row 1, so that the partner's bounded poll runs out (~0.2 s) -- the call must then
 *                 return HADI_ERR_INTERNAL instead of a field solved with stale exchange values; 128 = one block of every
 *                 team of an instance-resident launch deserts before the first team barrier -- the launch must time out,
 *                 and the call must still return the right field (from the two-launches-per-step path); 16 / 32 / 64 =
 *                 timing diagnostics of that launch (row phase / column phase / barriers skipped: results are wrong); 256 / 512 =
 *                 timing diagnostics of the column pass (tiles loaded and stored but not solved / solved without the reduced
 *                 system: results are wrong; tools/colpass_ab.py) */
int hadi_set_tuning(hadi_ctx *ctx, const char *key, int value);
int hadi_get_tuning(const hadi_ctx *ctx, const char *key, int *value);
/* Device the handle runs on: name, CU count, gcn arch string (for bench reports). */
int hadi_device_info(const hadi_ctx *ctx, char *name, int name_len, int *compute_units, char *arch, int arch_len);
/* Which kernels the last sweep on this handle ran (kernel names and tile geometry), for bench reports and profiles. */
int hadi_describe_last_sweep(const hadi_ctx *ctx, char *buf, int len);
/* Opaque hipStream_t of the handle (void*), so callers can order their own work after it. */
void *hadi_stream(hadi_ctx *ctx);
/* Input ordering (HADI_MEM_DEVICE).  The handle runs on its own non-blocking stream, which does not wait for the
 * caller's streams: work the caller enqueued on `producer_stream` (a hipStream_t; NULL = the legacy default stream)
 * that writes arrays of the next hadi_* call must be ordered before it with this call (an event is recorded on
 * `producer_stream` and the handle's stream waits for it; nothing blocks on the host), or the caller must synchronise
 * that stream itself.  The reference gets the same guarantee from deep_copy + Kokkos::fence (device_solver.cpp:718-726).
 * Outputs need no such call: every hadi_* entry point returns after its results are complete. */
int hadi_wait_stream(hadi_ctx *ctx, void *producer_stream);

/* ---- grids (host code, no GPU needed) ------------------------------------------------------ */
int hadi_make_grid(int m1, double S, double S_0, double K, double c,
                   int m2, double V, double V_0, double d,
                   double *vec_s, double *vec_v, double *delta_s, double *delta_v);
int hadi_rebuild_variance(int m2, double V_0_new, double V, double d, double *vec_v, double *delta_v);
int hadi_find_s_index(int m1, const double *vec_s, double S_0);  /* -1 if not a node */
int hadi_find_v_index(int m2, const double *vec_v, double V_0);  /* 0 if not a node (grid_pod.hpp:76-87) */

/* ---- the hot path -------------------------------------------------------------------------- */
int hadi_DO_timestepping(hadi_ctx *ctx, const hadi_problem *p);

int hadi_parallel_DO_solve(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0,
                           double *base_prices /* [n] */);

/* v-grid rebuilt from (V_0, V = 5.0, d = 5.0/500) exactly as every call site of the reference
 * does (jacobian_computation.cpp:253); p->vec_v / p->delta_v are ignored. */
int hadi_compute_base_prices(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0,
                             double *base_prices);
int hadi_compute_base_prices_american(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0,
                                      double *base_prices);
int hadi_compute_base_prices_dividends(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0,
                                       double *base_prices);
int hadi_compute_base_prices_american_dividends(hadi_ctx *ctx, const hadi_problem *p, double S_0,
                                                double V_0, double *base_prices);

/* J: [n][5], columns kappa, eta, sigma, rho, v0; forward differences with step eps.
 * p->U is not used as input (the reference restarts every solve from U_0); p->U_0 is required. */
int hadi_compute_jacobian(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0, double eps,
                          double *J, double *base_prices);
int hadi_compute_jacobian_american(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0,
                                   double eps, double *J, double *base_prices);
int hadi_compute_jacobian_dividends(hadi_ctx *ctx, const hadi_problem *p, double S_0, double V_0,
                                    double eps, double *J, double *base_prices);
int hadi_compute_jacobian_american_dividends(hadi_ctx *ctx, const hadi_problem *p, double S_0,
                                             double V_0, double eps, double *J, double *base_prices);

/* Levenberg-Marquardt normal equations on this rank's rows (host arrays):
 * partial[0..24] = J^T J (row-major), partial[25..29] = J^T r, partial[30] = sum r^2.
 * The 31 doubles are what gets all-reduced across GPUs. */
int hadi_lm_partials(int n, const double *J, const double *residuals, double *partial31);
/* The same reduction on the GPU (replaces KokkosBlas gemm/gemv + the residual kernel, src/jacobian_computation.cpp:117,
 * 154, heston_calibration.cpp:271-275): J [n][5], model prices [n] and market prices [n] are DEVICE arrays (what
 * hadi_compute_jacobian wrote with HADI_MEM_DEVICE); r = market - model; the 31 doubles come back to the host array
 * partial31 -- the only data that leaves the GPU per LM iteration.  Deterministic (fixed-shape tree reduction). */
int hadi_lm_partials_device(hadi_ctx *ctx, int n, const double *J, const double *model_prices,
                            const double *market_prices, double *partial31);
/* delta = (J^T J with diagonal *(1+lambda))^-1 J^T r by 5x5 partial-pivot elimination. */
int hadi_lm_solve(const double *partial31, double lambda, double *delta5);
/* Both steps on one rank: compute_parameter_update_on_device. */
int hadi_compute_parameter_update(int n, const double *J, const double *residuals, double lambda,
                                  double *delta5);


/* ---- diagnostics (tests): the two directional passes of ONE Douglas step as operators ----------------------------
 * hadi_debug_row_pass:   Y1rhs = what the row pass of step `step` hands to the A2 solve (device_solver.hpp:236-260:
 *                        explicit stage, A1 line solve, + theta dt (b2 (e_n - e_{n-1}) - A2 U)), from p->U as U_{n-1}.
 * hadi_debug_col_solve:  X = (I - theta dt A2)^{-1} p->U  (hes_a2_shuffled_kernels.hpp:243-299), European, no projection.
 * Both run the product kernels the batch shape selects; p->U is not modified; out is [n][m] in p->memspace. */
int hadi_debug_row_pass(hadi_ctx *ctx, const hadi_problem *p, int step, double *Y1rhs);
int hadi_debug_col_solve(hadi_ctx *ctx, const hadi_problem *p, double *X);
/* out[k] = the reciprocal of x[k] exactly as the line solves of the sweep form it (v_rcp_f64 + one Newton step: within
 * 10 ulp, no IEEE division); x and out are HOST arrays of n doubles.  Tests bound its error. */
int hadi_debug_rcp(hadi_ctx *ctx, int n, const double *x, double *out);

#ifdef __cplusplus
}
#endif
#endif /* HADI_H */
