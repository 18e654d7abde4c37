/*
 * heston_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64, sequential per instance) of the batched
 * Douglas-ADI Heston time-stepper of BCW-dot/PDE-based-Heston-Solver-GPU-accelerated.
 * It is the checker for the HIP path and the "port" CPU baseline in bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product library (libhadi) never links or calls it.
 *
 * Parity pin: see oracle/README.md -- the reference needs Kokkos and cannot be
 * built here (no oracle/_ref).  The oracle is pinned (A) against data the reference
 * holds in its own sources: the test_convergence sweep and its semi-analytic target
 * (src/solver.cpp:1653-1692), the hard-coded print targets, the per-operator
 * acceptance drivers (tests/test_reference_pins.py); and (B) at price / Jacobian /
 * calibration level against the reference outputs recorded in SURVEY.md section 8(c)
 * (tests/golden/, survey-recorded: that build cannot be repeated).  Full fields, the
 * put boundary data and the fp32 state have no reference counterpart: unpinned,
 * validated on their own (oracle/README.md).
 */
#ifndef HESTON_ORACLE_H
#define HESTON_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { HO_EU = 0, HO_AM = 1, HO_DIV = 2, HO_AM_DIV = 3 };

typedef struct ho_params {
    int m1, m2, N;
    int variant;
    double delta_t, theta;
    double r_d, r_f;
    double rho, sigma, kappa, eta;
    int num_dividends;
    const double *div_dates, *div_amounts, *div_percentages;
    int scheme; /* 0 = Douglas (device_solver.hpp), 1 = Craig-Sneyd (solver.hpp:781-907, European only) */
    int state_fp32; /* 1 = round the state to float between the directional passes (checker for libhadi's fp32-state
                     * sweep; NOT a reference feature, European Douglas only) */
    /* Put boundary data (checker for libhadi's HADI_PUT; NOT a reference feature -- the reference's only boundary class
     * is call-specific, BoundaryConditions.hpp:7-12).  option_type 1: b1 == 0 (du/ds = 0 at s_max), b2 = -1/2 r_d K on
     * the last v-row with time factor e^{-r_d dt n} (u = K e^{-r_d t} at v_max), and the i = 0 row of A1 carries the
     * reaction term -1/2 r_d (u = K e^{-r_d t} at s = 0).  `strike` is K; batch calls read strike_i[k] when non-NULL. */
    int option_type;
    double strike;
    const double *strike_i;
} ho_params;

/* Optional capture of the intermediates of time step `step` (1-based); every
 * non-NULL pointer receives m = (m1+1)(m2+1) doubles. */
typedef struct ho_dump {
    int step;
    double *b, *b1, *b2;
    double *A0U, *A1U, *A2U;
    double *Y0rhs, *Y1, *Y1rhs, *Unext;
} ho_dump;

/* grid.cpp:16-61 */
void ho_grid(int m1, double S, double S_0, double K, double c,
             int m2, double V, double V_0, double d,
             double *vec_s, double *vec_v, double *delta_s, double *delta_v);
/* grid_pod.hpp:25-73 */
void ho_rebuild_variance(int m2, double V_0_new, double V, double d,
                         double *vec_v, double *delta_v);
/* jacobian_computation.cpp:275-281 (-1 if S_0 is no node) */
int ho_find_s_index(int m1, const double *vec_s, double S_0);
/* grid_pod.hpp:76-87 (0 if V_0 is no node) */
int ho_find_v_index(int m2, const double *vec_v, double V_0);

/* device_DO_timestepping{,_american,_dividend,_american_dividend}
 * (device_solver.hpp:194-942) preceded by Device_BoundaryConditions::initialize
 * and the three build_matrix calls.  U: in = initial condition, out = solution.
 * U_0 (payoff) and lambda_bar (m doubles scratch/out) are needed for AM variants. */
int ho_solve(const ho_params *p,
             const double *vec_s, const double *vec_v,
             const double *delta_s, const double *delta_v,
             double *U, const double *U_0, double *lambda_bar, ho_dump *dump);

/* n independent instances, OpenMP over instances (Kokkos TeamPolicy(n, AUTO)
 * on a host backend).  Arrays are [n][...] row-major.  Returns threads used. */
int ho_solve_batch(const ho_params *p, int n,
                   const double *vec_s, const double *vec_v,
                   const double *delta_s, const double *delta_v,
                   double *U, const double *U_0, double *lambda_bar, int threads);

/* compute_base_prices* (jacobian_computation.cpp:368-448 etc.): rebuilds the
 * v-grid from (V_0, V, d) IN PLACE, solves from U, picks the price. */
int ho_base_prices(const ho_params *p, int n, double S_0, double V_0, double V, double d,
                   const double *vec_s, double *vec_v,
                   const double *delta_s, double *delta_v,
                   double *U, const double *U_0, double *base_prices, int threads);

/* compute_jacobian* (jacobian_computation.cpp:204-364 etc.): columns
 * kappa, eta, sigma, rho, v0; forward differences with step eps.  Leaves the
 * v-grid rebuilt for V_0+eps, as the reference does. */
int ho_jacobian(const ho_params *p, int n, double S_0, double V_0, double V, double d,
                const double *vec_s, double *vec_v,
                const double *delta_s, double *delta_v,
                const double *U_0, double *J, double *base_prices, double eps, int threads);

/* solve_5x5_device + compute_parameter_update_on_device
 * (jacobian_computation.cpp:20-195): delta = (J^T J (.) (1+lambda on diag))^-1 J^T r */
void ho_lm_update(int n, const double *J, const double *residuals, double lambda, double *delta);

/* The reference's per-operator acceptance drivers (hes_a0_kernels.cpp:8-121, hes_a1_kernels.cpp:130-277,
 * hes_a2_shuffled_kernels.cpp:13-157,271-277): build the three operators for one instance, then
 *   which = 0: result = A0 x                              (multiply_parallel_v)
 *   which = 1: result = A1 x,  xsol = (I - theta dt A1)^{-1} b   (multiply_parallel_v, solve_implicit_parallel_v)
 *   which = 2: result = A2 x,  xsol = (I - theta dt A2)^{-1} b   (multiply_parallel_s, solve_implicit_parallel_s;
 *              x, b, result, xsol in the NATURAL layout idx = i + j (m1+1): the shuffle is internal)
 * so that a test can form the printed residual ||xsol - theta dt A xsol - b||_2.  result / xsol may be NULL. */
int ho_operator(const ho_params *p, int which,
                const double *vec_s, const double *vec_v, const double *delta_s, const double *delta_v,
                const double *x, const double *b, double *result, double *xsol);

int ho_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
