/*
 * heston_oracle_xp.c -- TEST INFRASTRUCTURE ONLY: the adjudicator.
 *
 * The UNMODIFIED oracle source (heston_oracle.c, the restatement of the reference's device family) compiled a
 * second time with every `double` replaced by IEEE binary128 (__float128, libquadmath; -DHOX_LONG_DOUBLE gives
 * the x87 80-bit type instead).  Where libhadi and the fp64 oracle differ by more than their usual 1e-12 -- both
 * run the same scheme in fp64 with different operation orders, and an s-grid with a 0.04-wide interval beside a
 * 5-wide one amplifies round-off by ~1e9 -- this build says which of the two is closer to the scheme's exact
 * result for the SAME fp64 inputs (grids, increments, parameters, payoff are handed over as doubles and widened
 * exactly; outputs are rounded to double once, at the end).
 *
 * Nothing of the algorithm lives here: only the type switch and a flat fp64 entry point for one instance.
 * Only tests/ and tools/fuzz_parity.py load the resulting library (oracle.py: solve_xp).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double hox_f64; /* the real fp64, before `double` changes its meaning */

#ifdef HOX_LONG_DOUBLE
typedef long double hox_real;
#define exp expl
#define sinh sinhl
#define asinh asinhl
#define fabs fabsl
#define fmax fmaxl
#else
#include <quadmath.h>
typedef __float128 hox_real;
#define exp expq
#define sinh sinhq
#define asinh asinhq
#define fabs fabsq
#define fmax fmaxq
#endif

/* Dividend dating compares n*dt with the dates in FLOATING POINT and the reference's fp64 round-off decides the
 * step a dividend lands on (12*0.05 = 0.6000000000000001): that decision stays fp64 in this build too. */
#define HO_F64 hox_f64

/* every public name of the oracle gets its own name here, so both libraries can be loaded side by side */
#define ho_params hox_params
#define ho_dump hox_dump
#define ho_grid hox_grid
#define ho_rebuild_variance hox_rebuild_variance
#define ho_find_s_index hox_find_s_index
#define ho_find_v_index hox_find_v_index
#define ho_solve hox_solve
#define ho_operator hox_operator
#define ho_max_threads hox_max_threads
#define ho_solve_batch hox_solve_batch
#define ho_base_prices hox_base_prices
#define ho_jacobian hox_jacobian
#define ho_lm_update hox_lm_update

#define double hox_real
#include "heston_oracle.c"
#undef double

static hox_real *widen(const hox_f64 *a, size_t n) {
    if (!a) return NULL;
    hox_real *r = (hox_real *)malloc((n ? n : 1) * sizeof(hox_real));
    for (size_t i = 0; i < n; i++) r[i] = (hox_real)a[i];
    return r;
}

/* One instance, fp64 in / fp64 out, extended precision inside.  Arguments = the fields of ho_params + the arrays
 * of ho_solve.  U: in = initial condition, out = solution at T; lambda_out (American variants) may be NULL.
 * Returns ho_solve's status. */
int hoxp_solve(int m1, int m2, int N, int variant, hox_f64 delta_t, hox_f64 theta, hox_f64 r_d, hox_f64 r_f,
               hox_f64 rho, hox_f64 sigma, hox_f64 kappa, hox_f64 eta,
               int num_dividends, const hox_f64 *div_dates, const hox_f64 *div_amounts, const hox_f64 *div_percentages,
               int scheme, int state_fp32, int option_type, hox_f64 strike,
               const hox_f64 *vec_s, const hox_f64 *vec_v, const hox_f64 *delta_s, const hox_f64 *delta_v,
               hox_f64 *U, const hox_f64 *U_0, hox_f64 *lambda_out) {
    const size_t m = (size_t)(m1 + 1) * (m2 + 1);
    hox_params p;
    memset(&p, 0, sizeof p);
    p.m1 = m1; p.m2 = m2; p.N = N; p.variant = variant;
    p.delta_t = delta_t; p.theta = theta; p.r_d = r_d; p.r_f = r_f;
    p.rho = rho; p.sigma = sigma; p.kappa = kappa; p.eta = eta;
    p.num_dividends = num_dividends;
    hox_real *dd = widen(div_dates, num_dividends), *da = widen(div_amounts, num_dividends), *dp = widen(div_percentages, num_dividends);
    p.div_dates = dd; p.div_amounts = da; p.div_percentages = dp;
    p.scheme = scheme; p.state_fp32 = state_fp32; p.option_type = option_type; p.strike = strike; p.strike_i = NULL;
    hox_real *vs = widen(vec_s, m1 + 1), *vv = widen(vec_v, m2 + 1), *ds = widen(delta_s, m1), *dv = widen(delta_v, m2);
    hox_real *Ux = widen(U, m), *U0x = widen(U_0, m);
    hox_real *lam = (hox_real *)calloc(m, sizeof(hox_real));
    const int rc = hox_solve(&p, vs, vv, ds, dv, Ux, U0x, lam, NULL);
    if (rc == 0) {
        for (size_t i = 0; i < m; i++) U[i] = (hox_f64)Ux[i];
        if (lambda_out)
            for (size_t i = 0; i < m; i++) lambda_out[i] = (hox_f64)lam[i];
    }
    free(dd); free(da); free(dp); free(vs); free(vv); free(ds); free(dv); free(Ux); free(U0x); free(lam);
    return rc;
}

/* significant bits of the arithmetic type of this build (113: binary128, 64: x87 extended) */
int hoxp_mantissa_bits(void) {
#ifdef HOX_LONG_DOUBLE
    return 64;
#else
    return 113;
#endif
}
