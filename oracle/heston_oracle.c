/*
 * heston_oracle.c -- TEST INFRASTRUCTURE ONLY (see heston_oracle.h).
 *
 * Plain-C restatement of the reference's *device-callable* operator family and
 * Douglas time steppers, written from the arithmetic spec in SURVEY.md
 * Appendix A and the cited reference lines.  Every quirk that is "in the
 * reference's numbers" is reproduced on purpose and marked (quirk).
 *
 * Storage differs from the reference only where the stored numbers are
 * provably identical: A2 diagonals do not depend on the s-index
 * (hes_a2_shuffled_kernels.hpp:122-152) and are kept once; the shuffle /
 * unshuffle transposes (hes_a2_shuffled_kernels.hpp:12-44) are pure data
 * movement and are replaced by strided indexing.  The order of every
 * floating-point operation follows the reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 */
#include "heston_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* fp64 under its own name: the adjudicator build (heston_oracle_xp.c) compiles this file with `double` redefined to an
 * extended type, but the dividend dating below must keep the reference's fp64 round-off either way. */
#ifndef HO_F64
#define HO_F64 double
#endif

/* ---- coeff.hpp:24-126 : 3-point non-uniform FD weights ------------------ */
static double fd_delta(const double *D, int i, int pos) {
    if (pos == -1) return 2 / (D[i] * (D[i] + D[i + 1]));
    if (pos == 0) return -2 / (D[i] * D[i + 1]);
    if (pos == 1) return 2 / (D[i + 1] * (D[i] + D[i + 1]));
    return 0.0;
}
static double fd_beta(const double *D, int i, int pos) {
    if (pos == -1) return -D[i + 1] / (D[i] * (D[i] + D[i + 1]));
    if (pos == 0) return (D[i + 1] - D[i]) / (D[i] * D[i + 1]);
    if (pos == 1) return D[i] / (D[i + 1] * (D[i] + D[i + 1]));
    return 0.0;
}
static double fd_alpha(const double *D, int i, int pos) {
    if (pos == -2) return D[i] / (D[i - 1] * (D[i - 1] + D[i]));
    if (pos == -1) return (-D[i - 1] - D[i]) / (D[i - 1] * D[i]);
    if (pos == 0) return (D[i - 1] + 2 * D[i]) / (D[i] * (D[i - 1] + D[i]));
    return 0.0;
}
static double fd_gamma(const double *D, int i, int pos) {
    if (pos == 0) return (-2 * D[i + 1] - D[i + 2]) / (D[i + 1] * (D[i + 1] + D[i + 2]));
    if (pos == 1) return (D[i + 1] + D[i + 2]) / (D[i + 1] * D[i + 2]);
    if (pos == 2) return -D[i + 1] / (D[i + 2] * (D[i + 1] + D[i + 2]));
    return 0.0;
}

/* ---- grid.cpp:16-61 ------------------------------------------------------ */
/* push_back(x0); std::sort; pop_back  ==  sorted insert, drop the largest. */
static void insert_sorted_drop_last(double *v, int n, double x0) {
    /* v has n ascending entries; result keeps n entries. */
    if (x0 >= v[n - 1]) return; /* x0 would be (one of) the largest: dropped */
    int pos = 0;
    while (pos < n && v[pos] <= x0) pos++;
    for (int k = n - 1; k > pos; k--) v[k] = v[k - 1];
    v[pos] = x0;
}

void ho_grid(int m1, double S, double S_0, double K, double c,
             int m2, double V, double V_0, double d,
             double *vec_s, double *vec_v, double *delta_s, double *delta_v) {
    const double Delta_xi = (1.0 / m1) * (asinh((S - K) / c) - asinh(-K / c));
    for (int i = 0; i <= m1; i++) {
        const double xi = asinh(-K / c) + i * Delta_xi;
        vec_s[i] = K + c * sinh(xi);
    }
    insert_sorted_drop_last(vec_s, m1 + 1, S_0);
    for (int i = 0; i < m1; i++) delta_s[i] = vec_s[i + 1] - vec_s[i];

    const double Delta_eta = (1.0 / m2) * asinh(V / d);
    for (int i = 0; i <= m2; i++) {
        const double xi = i * Delta_eta;
        vec_v[i] = d * sinh(xi);
    }
    insert_sorted_drop_last(vec_v, m2 + 1, V_0);
    for (int i = 0; i < m2; i++) delta_v[i] = vec_v[i + 1] - vec_v[i];
}

/* ---- grid_pod.hpp:25-73 (bubble sort of m2+2 values == sorted insert) ---- */
void ho_rebuild_variance(int m2, double V_0_new, double V, double d,
                         double *vec_v, double *delta_v) {
    const double Delta_eta = (1.0 / m2) * asinh(V / d);
    for (int i = 0; i <= m2; i++) {
        const double xi = i * Delta_eta;
        vec_v[i] = d * sinh(xi);
    }
    insert_sorted_drop_last(vec_v, m2 + 1, V_0_new);
    for (int i = 0; i < m2; i++) delta_v[i] = vec_v[i + 1] - vec_v[i];
}

int ho_find_s_index(int m1, const double *vec_s, double S_0) {
    for (int i = 0; i <= m1; i++)
        if (fabs(vec_s[i] - S_0) < 1e-10) return i;
    return -1;
}
int ho_find_v_index(int m2, const double *vec_v, double V_0) {
    for (int i = 0; i <= m2; i++)
        if (fabs(vec_v[i] - V_0) < 1e-10) return i;
    return 0; /* (quirk) grid_pod.hpp:78,86 */
}

/* ---- operator storage ----------------------------------------------------- */
typedef struct {
    int m1, m2, m;
    /* A0: hes_a0_kernels.hpp:22 values[(m2-1)][9(m1-1)] */
    double *a0;
    /* A1: hes_a1_kernels.hpp:21-30, rows j=0..m2 */
    double *a1_main, *a1_lower, *a1_upper;       /* [(m2+1)][(m1+1)] (lower/upper use m1 of them) */
    double *a1_imain, *a1_ilower, *a1_iupper, *a1_tmp;
    /* A2: hes_a2_shuffled_kernels.hpp:58-75, identical for every i -> one copy */
    double *a2_main, *a2_lower, *a2_lower2, *a2_upper, *a2_upper2;
    double *a2_imain, *a2_ilower, *a2_ilower2, *a2_iupper, *a2_iupper2;
    double *a2_c, *a2_c2, *a2_q, *a2_L; /* factorisation (c_prime, c2_prime, 1/den, l - l2 c) */
    /* boundary vectors hes_boundary_kernels.hpp:11-14 */
    double *b0, *b1, *b2, *b;
    /* work */
    double *A0U, *A1U, *A2U, *Y0, *Y1, *Utmp, *dcol;
} ho_ws;

static double *zalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

static ho_ws *ws_new(int m1, int m2) {
    ho_ws *w = (ho_ws *)calloc(1, sizeof(ho_ws));
    const size_t m = (size_t)(m1 + 1) * (m2 + 1);
    w->m1 = m1; w->m2 = m2; w->m = (int)m;
    w->a0 = zalloc((size_t)(m2 > 1 ? m2 - 1 : 0) * 9 * (m1 > 1 ? m1 - 1 : 0));
    w->a1_main = zalloc(m); w->a1_lower = zalloc(m); w->a1_upper = zalloc(m);
    w->a1_imain = zalloc(m); w->a1_ilower = zalloc(m); w->a1_iupper = zalloc(m); w->a1_tmp = zalloc(m);
    const size_t n2 = (size_t)m2 + 2;
    w->a2_main = zalloc(n2); w->a2_lower = zalloc(n2); w->a2_lower2 = zalloc(n2);
    w->a2_upper = zalloc(n2); w->a2_upper2 = zalloc(n2);
    w->a2_imain = zalloc(n2); w->a2_ilower = zalloc(n2); w->a2_ilower2 = zalloc(n2);
    w->a2_iupper = zalloc(n2); w->a2_iupper2 = zalloc(n2);
    w->a2_c = zalloc(n2); w->a2_c2 = zalloc(n2); w->a2_q = zalloc(n2); w->a2_L = zalloc(n2);
    w->b0 = zalloc(m); w->b1 = zalloc(m); w->b2 = zalloc(m); w->b = zalloc(m);
    w->A0U = zalloc(m); w->A1U = zalloc(m); w->A2U = zalloc(m);
    w->Y0 = zalloc(m); w->Y1 = zalloc(m); w->Utmp = zalloc(m); w->dcol = zalloc(n2);
    return w;
}
static void ws_free(ho_ws *w) {
    if (!w) return;
    free(w->a0);
    free(w->a1_main); free(w->a1_lower); free(w->a1_upper);
    free(w->a1_imain); free(w->a1_ilower); free(w->a1_iupper); free(w->a1_tmp);
    free(w->a2_main); free(w->a2_lower); free(w->a2_lower2); free(w->a2_upper); free(w->a2_upper2);
    free(w->a2_imain); free(w->a2_ilower); free(w->a2_ilower2); free(w->a2_iupper); free(w->a2_iupper2);
    free(w->a2_c); free(w->a2_c2); free(w->a2_q); free(w->a2_L);
    free(w->b0); free(w->b1); free(w->b2); free(w->b);
    free(w->A0U); free(w->A1U); free(w->A2U); free(w->Y0); free(w->Y1); free(w->Utmp); free(w->dcol);
    free(w);
}

/* ---- hes_boundary_kernels.hpp:41-75 --------------------------------------- */
static void bc_initialize(ho_ws *w, const double *vec_s, double r_d, double r_f, int N, double dt, int put, double strike) {
    const int m1 = w->m1, m2 = w->m2, m = w->m;
    memset(w->b0, 0, sizeof(double) * m);
    memset(w->b1, 0, sizeof(double) * m);
    memset(w->b2, 0, sizeof(double) * m);
    memset(w->b, 0, sizeof(double) * m);
    if (put) { /* NOT in the reference (see ho_params.option_type): b1 == 0; b2 makes u = K e^{-r_d t} exact on the last
                * v-row; the time factor e^{-r_d dt n} is applied by timestepping() */
        for (int i = 0; i <= m1; i++) w->b2[m - m1 - 1 + i] = -0.5 * r_d * strike;
        for (int i = 0; i < m; i++) w->b[i] = w->b0[i] + w->b1[i] + w->b2[i];
        return;
    }
    for (int j = 0; j <= m2; j++) {
        const double exp_factor = exp(-r_f * dt * (N - 1));
        const long idx = (long)m1 * (j + 1); /* (quirk) not m1 + j*(m1+1) */
        if (idx < m) w->b1[idx] = (r_d - r_f) * vec_s[m1] * exp_factor;
    }
    for (int i = 0; i <= m1; i++) {
        const double exp_factor = exp(-r_f * dt * (N - 1));
        w->b2[m - m1 - 1 + i] = -0.5 * r_d * vec_s[i] * exp_factor;
    }
    for (int i = 0; i < m; i++) w->b[i] = w->b0[i] + w->b1[i] + w->b2[i];
}

/* ---- hes_a0_kernels.hpp:30-56 --------------------------------------------- */
static void a0_build(ho_ws *w, const double *vec_s, const double *vec_v,
                     const double *ds, const double *dv, double rho, double sigma) {
    const int m1 = w->m1, m2 = w->m2;
    for (int j = 0; j < m2 - 1; j++) {
        double *row = w->a0 + (size_t)j * 9 * (m1 - 1);
        for (int i = 0; i < m1 - 1; i++) {
            const double c = rho * sigma * vec_s[i + 1] * vec_v[j + 1];
            for (int l = -1; l <= 1; l++)
                for (int k = -1; k <= 1; k++) {
                    const int val_idx = i * 9 + (l + 1) * 3 + (k + 1);
                    const double beta_s_val = fd_beta(ds, i, k);
                    const double beta_v_val = fd_beta(dv, j, l);
                    row[val_idx] = c * beta_s_val * beta_v_val;
                }
        }
    }
}
/* hes_a0_kernels.hpp:59-94 */
static void a0_multiply(const ho_ws *w, const double *x, double *result) {
    const int m1 = w->m1, m2 = w->m2, m = w->m;
    for (int i = 0; i < m; i++) result[i] = 0.0;
    for (int j = 0; j < m2 - 1; j++) {
        const double *row = w->a0 + (size_t)j * 9 * (m1 - 1);
        for (int i = 0; i < m1 - 1; i++) {
            const int row_offset = (j + 1) * (m1 + 1) + (i + 1);
            double sum = 0.0;
            for (int l = -1; l <= 1; l++)
                for (int k = -1; k <= 1; k++) {
                    const int val_idx = i * 9 + (l + 1) * 3 + (k + 1);
                    const int col_idx = (i + 1 + k) + (j + 1 + l) * (m1 + 1);
                    if (col_idx >= 0 && col_idx < m) sum += row[val_idx] * x[col_idx];
                }
            result[row_offset] = sum;
        }
    }
}

/* ---- hes_a1_kernels.hpp:51-107 -------------------------------------------- */
static void a1_build(ho_ws *w, const double *vec_s, const double *vec_v, const double *ds,
                     double r_d, double r_f, double theta, double dt, int put) {
    const int m1 = w->m1, m2 = w->m2, ld = m1 + 1;
    for (int j = 0; j <= m2; j++) {
        double *mn = w->a1_main + (size_t)j * ld, *lo = w->a1_lower + (size_t)j * ld, *up = w->a1_upper + (size_t)j * ld;
        double *imn = w->a1_imain + (size_t)j * ld, *ilo = w->a1_ilower + (size_t)j * ld, *iup = w->a1_iupper + (size_t)j * ld;
        mn[0] = 0.0; imn[0] = 1.0;
        up[0] = 0.0; iup[0] = 0.0; /* reference guards j<m2; row m2 stays View-zero: same value */
        if (put) { /* NOT in the reference: the s = 0 row carries the reaction term, so that u(0) = K e^{-r_d t} */
            mn[0] = -0.5 * r_d;
            imn[0] = 1.0 - theta * dt * mn[0];
        }
        for (int i = 1; i < m1; i++) {
            const double s = vec_s[i];
            const double v = vec_v[j];
            const double a = 0.5 * s * s * v;
            const double b = (r_d - r_f) * s;
            lo[i - 1] = a * fd_delta(ds, i - 1, -1) + b * fd_beta(ds, i - 1, -1);
            mn[i] = a * fd_delta(ds, i - 1, 0) + b * fd_beta(ds, i - 1, 0) - 0.5 * r_d;
            up[i] = a * fd_delta(ds, i - 1, 1) + b * fd_beta(ds, i - 1, 1);
            ilo[i - 1] = -theta * dt * lo[i - 1];
            imn[i] = 1.0 - theta * dt * mn[i];
            iup[i] = -theta * dt * up[i];
        }
        mn[m1] = -0.5 * r_d;
        imn[m1] = 1.0 - theta * dt * mn[m1];
        lo[m1 - 1] = 0.0;
        ilo[m1 - 1] = 0.0;
    }
}
/* hes_a1_kernels.hpp:111-135 */
static void a1_multiply(const ho_ws *w, const double *x, double *result) {
    const int m1 = w->m1, m2 = w->m2, ld = m1 + 1;
    for (int j = 0; j <= m2; j++) {
        const double *mn = w->a1_main + (size_t)j * ld, *lo = w->a1_lower + (size_t)j * ld, *up = w->a1_upper + (size_t)j * ld;
        const int offset = j * ld;
        double sum = mn[0] * x[offset];
        sum += up[0] * x[offset + 1];
        result[offset] = sum;
        for (int i = 1; i < m1; i++) {
            double s2 = lo[i - 1] * x[offset + i - 1] + mn[i] * x[offset + i] + up[i] * x[offset + i + 1];
            result[offset + i] = s2;
        }
        sum = lo[m1 - 1] * x[offset + m1 - 1] + mn[m1] * x[offset + m1];
        result[offset + m1] = sum;
    }
}
/* hes_a1_kernels.hpp:139-161 */
static void a1_solve(ho_ws *w, double *x, const double *b) {
    const int m1 = w->m1, m2 = w->m2, ld = m1 + 1;
    for (int j = 0; j <= m2; j++) {
        const double *imn = w->a1_imain + (size_t)j * ld, *ilo = w->a1_ilower + (size_t)j * ld, *iup = w->a1_iupper + (size_t)j * ld;
        double *tp = w->a1_tmp + (size_t)j * ld;
        const int offset = j * ld;
        tp[0] = imn[0];
        x[offset] = b[offset];
        for (int i = 1; i <= m1; i++) {
            const double mm = ilo[i - 1] / tp[i - 1];
            tp[i] = imn[i] - mm * iup[i - 1];
            x[offset + i] = b[offset + i] - mm * x[offset + i - 1];
        }
        x[offset + m1] /= tp[m1];
        for (int i = m1 - 1; i >= 0; i--)
            x[offset + i] = (x[offset + i] - iup[i] * x[offset + i + 1]) / tp[i];
    }
}

/* ---- hes_a2_shuffled_kernels.hpp:103-176 ----------------------------------- */
static void a2_build(ho_ws *w, const double *vec_v, const double *dv,
                     double r_d, double kappa, double eta, double sigma, double theta, double dt) {
    const int m2 = w->m2;
    double *mn = w->a2_main, *lo = w->a2_lower, *lo2 = w->a2_lower2, *up = w->a2_upper, *up2 = w->a2_upper2;
    for (int j = 0; j < m2 + 2; j++) { mn[j] = lo[j] = lo2[j] = up[j] = up2[j] = 0.0; }
    for (int j = 0; j < m2 - 1; j++) {
        const double temp = kappa * (eta - vec_v[j]);
        const double temp2 = 0.5 * sigma * sigma * vec_v[j];
        mn[j] += -0.5 * r_d;
        if (vec_v[j] > 1.0) { /* (quirk) targets row j+1 */
            lo2[j + 1 - 2] += temp * fd_alpha(dv, j, -2);
            lo[j + 1 - 1] += temp * fd_alpha(dv, j, -1);
            mn[j + 1 - 0] += temp * fd_alpha(dv, j, 0);
            lo[j + 1 - 1] += temp2 * fd_delta(dv, j - 1, -1);
            mn[j + 1 + 0] += temp2 * fd_delta(dv, j - 1, 0);
            up[j + 1] += temp2 * fd_delta(dv, j - 1, 1);
        }
        if (j == 0) {
            mn[j] += temp * fd_gamma(dv, j, 0);
            up[j] += temp * fd_gamma(dv, j, 1);
            up2[j] += temp * fd_gamma(dv, j, 2);
        } else {
            lo[j - 1] += temp * fd_beta(dv, j - 1, -1) + temp2 * fd_delta(dv, j - 1, -1);
            mn[j] += temp * fd_beta(dv, j - 1, 0) + temp2 * fd_delta(dv, j - 1, 0);
            up[j] += temp * fd_beta(dv, j - 1, 1) + temp2 * fd_delta(dv, j - 1, 1);
        }
    }
    for (int j = 0; j <= m2; j++) w->a2_imain[j] = 1.0 - theta * dt * mn[j];
    for (int j = 0; j < m2; j++) {
        w->a2_ilower[j] = -theta * dt * lo[j];
        w->a2_iupper[j] = -theta * dt * up[j];
    }
    for (int j = 0; j < m2 - 1; j++) {
        w->a2_ilower2[j] = -theta * dt * lo2[j];
        w->a2_iupper2[j] = -theta * dt * up2[j];
    }
    /* entries past the reference's extents stay 0: the reference reads
     * impl_upper(i, m2) one past its row in the forward sweep (:272); the value
     * only feeds c_prime(i, m2), which back-substitution never uses. */
    w->a2_iupper[m2] = 0.0;

    /* Factorisation part of solve_implicit_parallel_s (:243-285); it does not
     * depend on the right-hand side nor on i, so it is evaluated once here with
     * the reference's expressions. */
    const int n = m2 + 1;
    double *c = w->a2_c, *c2 = w->a2_c2, *q = w->a2_q, *L = w->a2_L;
    const double *d = w->a2_imain, *l = w->a2_ilower, *l2 = w->a2_ilower2, *u = w->a2_iupper, *u2 = w->a2_iupper2;
    for (int j = 0; j < m2 + 2; j++) { c[j] = c2[j] = q[j] = L[j] = 0.0; }
    c[0] = u[0] / d[0];
    c2[0] = u2[0] / d[0];
    if (n > 1) {
        const double mm = 1.0 / (d[1] - l[0] * c[0]);
        c[1] = (u[1] - l[0] * c2[0]) * mm;
        c2[1] = u2[1] * mm;
        q[1] = mm;
    }
    for (int j = 2; j < n; j++) {
        const double den = d[j] - (l[j - 1] - l2[j - 2] * c[j - 2]) * c[j - 1] - l2[j - 2] * c2[j - 2];
        const double mm = 1.0 / den;
        c[j] = (u[j] - (l[j - 1] - l2[j - 2] * c[j - 2]) * c2[j - 1]) * mm;
        if (j < n - 2) c2[j] = u2[j] * mm; /* (quirk f) else stays 0 */
        q[j] = mm;
        L[j] = l[j - 1] - l2[j - 2] * c[j - 2];
    }
}
/* hes_a2_shuffled_kernels.hpp:180-239 on the natural (unshuffled) layout */
static void a2_multiply(const ho_ws *w, const double *x, double *result) {
    const int m1 = w->m1, m2 = w->m2, ld = m1 + 1;
    const double *mn = w->a2_main, *lo = w->a2_lower, *lo2 = w->a2_lower2, *up = w->a2_upper, *up2 = w->a2_upper2;
#define X(j) x[(size_t)(j) * ld + i]
#define R(j) result[(size_t)(j) * ld + i]
    for (int i = 0; i <= m1; i++) {
        R(0) = mn[0] * X(0);
        if (0 < m2) R(0) += up[0] * X(1);
        if (1 < m2) R(0) += up2[0] * X(2);
        if (0 < m2) {
            R(1) = lo[0] * X(0) + mn[1] * X(1);
            if (1 < m2) R(1) += up[1] * X(2);
            if (2 < m2) R(1) += up2[1] * X(3);
        }
        for (int j = 2; j < m2 - 1; j++) {
            R(j) = lo2[j - 2] * X(j - 2) + lo[j - 1] * X(j - 1) + mn[j] * X(j) + up[j] * X(j + 1);
            if (j < m2 - 2) R(j) += up2[j] * X(j + 2);
        }
        if (m2 > 2) {
            const int j = m2 - 1;
            R(j) = lo2[j - 2] * X(j - 2) + lo[j - 1] * X(j - 1) + mn[j] * X(j);
            if (j < m2) R(j) += up[j] * X(j + 1);
        }
        if (m2 > 1) {
            const int j = m2;
            R(j) = lo2[j - 2] * X(j - 2) + lo[j - 1] * X(j - 1) + mn[j] * X(j);
        }
    }
#undef X
#undef R
}
/* hes_a2_shuffled_kernels.hpp:243-299, right-hand-side part */
static void a2_solve(ho_ws *w, double *x, const double *b) {
    const int m1 = w->m1, m2 = w->m2, ld = m1 + 1, n = m2 + 1;
    const double *c = w->a2_c, *c2 = w->a2_c2, *q = w->a2_q, *L = w->a2_L;
    const double *d = w->a2_imain, *l = w->a2_ilower, *l2 = w->a2_ilower2;
    double *dp = w->dcol;
    for (int i = 0; i <= m1; i++) {
        dp[0] = b[i] / d[0];
        if (n > 1) dp[1] = (b[(size_t)ld + i] - l[0] * dp[0]) * q[1];
        for (int j = 2; j < n; j++)
            dp[j] = (b[(size_t)j * ld + i] - L[j] * dp[j - 1] - l2[j - 2] * dp[j - 2]) * q[j];
        x[(size_t)(n - 1) * ld + i] = dp[n - 1];
        if (n > 1) x[(size_t)(n - 2) * ld + i] = dp[n - 2] - c[n - 2] * x[(size_t)(n - 1) * ld + i];
        for (int j = n - 3; j >= 0; j--)
            x[(size_t)j * ld + i] = dp[j] - c[j] * x[(size_t)(j + 1) * ld + i] - c2[j] * x[(size_t)(j + 2) * ld + i];
    }
}

/* ---- device_solver.hpp:448-504 (dividend jump) ----------------------------- */
static void dividend_jump(ho_ws *w, const double *vec_s, double *U, double amount, double pct, int put) {
    const int m1 = w->m1, m2 = w->m2, m = w->m;
    memcpy(w->Utmp, U, sizeof(double) * m);
    for (int j = 0; j <= m2; j++) {
        const int offset = j * (m1 + 1);
        for (int i = 0; i <= m1; i++) {
            const double old_s = vec_s[i];
            const double new_s = old_s * (1.0 - pct) - amount;
            if (new_s > 0) {
                int idx = 0;
                for (int k = 0; k <= m1; k++)
                    if (vec_s[k] > new_s) { idx = k; break; }
                if (idx > 0 && idx < m1 + 1) {
                    const double s_low = vec_s[idx - 1], s_high = vec_s[idx];
                    const double weight = (new_s - s_low) / (s_high - s_low);
                    const double val_low = w->Utmp[offset + idx - 1], val_high = w->Utmp[offset + idx];
                    U[offset + i] = (1.0 - weight) * val_low + weight * val_high;
                } else if (idx == 0) {
                    U[offset + i] = w->Utmp[offset];
                } else {
                    U[offset + i] = w->Utmp[offset + m1]; /* unreachable (quirk) */
                }
            } else {
                /* ex-dividend spot <= 0: a call is worth 0 (device_solver.hpp:499-503); put extension: its s = 0 value */
                U[offset + i] = put ? w->Utmp[offset] : 0.0;
            }
        }
    }
}

static void build_all(ho_ws *w, const ho_params *p, const double *vec_s, const double *vec_v,
                      const double *ds, const double *dv, double rho, double sigma, double kappa, double eta) {
    a0_build(w, vec_s, vec_v, ds, dv, rho, sigma);
    a1_build(w, vec_s, vec_v, ds, p->r_d, p->r_f, p->theta, p->delta_t, p->option_type == 1);
    a2_build(w, vec_v, dv, p->r_d, kappa, eta, sigma, p->theta, p->delta_t);
}

/* ---- solver.hpp:781-907 CS_scheme_shuffled (European): Douglas predictor, then a corrector that re-adds
 * half of the explicit mixed-derivative increment A0 Y2 - A0 U and repeats the two implicit solves.  The
 * reference runs it on its host operator family, which produces the same digits as the device family
 * (SURVEY.md 8(c)); here it reuses the device-family restatement above. */
static void cs_timestepping(ho_ws *w, const ho_params *p, double *U) {
    const int m = w->m, N = p->N;
    const double delta_t = p->delta_t, theta = p->theta, r_f = p->r_f;
    double *Y_0 = zalloc(m), *Y_1 = zalloc(m), *Y_2 = zalloc(m), *A0Y2 = zalloc(m), *Yt = zalloc(m);
    for (int n = 1; n <= N; n++) {
        a0_multiply(w, U, w->A0U);
        a1_multiply(w, U, w->A1U);
        a2_multiply(w, U, w->A2U);
        const double exp_factor_now = exp(r_f * delta_t * n);
        const double exp_factor_prev = exp(r_f * delta_t * (n - 1));
        for (int i = 0; i < m; i++)
            Y_0[i] = U[i] + delta_t * (w->A0U[i] + w->A1U[i] + w->A2U[i] + w->b[i] * exp_factor_prev);
        for (int i = 0; i < m; i++)
            Y_1[i] = Y_0[i] + theta * delta_t * (w->b1[i] * exp_factor_now - (w->A1U[i] + w->b1[i] * exp_factor_prev));
        a1_solve(w, Y_1, Y_1);
        for (int i = 0; i < m; i++)
            Y_2[i] = Y_1[i] + theta * delta_t * (w->b2[i] * exp_factor_now - (w->A2U[i] + w->b2[i] * exp_factor_prev));
        a2_solve(w, Y_2, Y_2);
        a0_multiply(w, Y_2, A0Y2);
        for (int i = 0; i < m; i++)
            Yt[i] = Y_0[i] + 0.5 * delta_t * ((A0Y2[i] + w->b0[i] * exp_factor_now) - (w->A0U[i] + w->b0[i] * exp_factor_prev));
        for (int i = 0; i < m; i++)
            Yt[i] = Yt[i] + theta * delta_t * (w->b1[i] * exp_factor_now - (w->A1U[i] + w->b1[i] * exp_factor_prev));
        a1_solve(w, Yt, Yt);
        for (int i = 0; i < m; i++)
            U[i] = Yt[i] + theta * delta_t * (w->b2[i] * exp_factor_now - (w->A2U[i] + w->b2[i] * exp_factor_prev));
        a2_solve(w, U, U);
    }
    free(Y_0); free(Y_1); free(Y_2); free(A0Y2); free(Yt);
}

/* ---- device_solver.hpp:194-942, all four variants ---------------------------- */
static void timestepping(ho_ws *w, const ho_params *p, const double *vec_s,
                         double *U, const double *U_0, double *lambda_bar, ho_dump *dump) {
    if (p->scheme == 1) {
        cs_timestepping(w, p, U);
        return;
    }
    const int m1 = w->m1, m = w->m, N = p->N;
    const double delta_t = p->delta_t, theta = p->theta;
    /* boundary data carry exp(rate dt n): rate = r_f in the reference (device_solver.hpp:238,246); -r_d for the put
     * extension (boundary value K e^{-r_d t}).  The variable keeps the reference's name. */
    const double r_f = (p->option_type == 1) ? -p->r_d : p->r_f;
    const int american = (p->variant == HO_AM || p->variant == HO_AM_DIV);
    const int dividend = (p->variant == HO_DIV || p->variant == HO_AM_DIV);
    int current_div_idx = 0;
    if (american)
        for (int i = 0; i < m; i++) lambda_bar[i] = 0;

    /* fp32-state sweep (libhadi's HADI_STATE_FP32; not a reference feature): the state is rounded to float where the
     * GPU stores it -- the initial U, the A2 right-hand side after the row pass, U after the column pass.  All
     * arithmetic stays fp64. */
    const int f32 = p->state_fp32;
    if (f32)
        for (int i = 0; i < m; i++) U[i] = (double)(float)U[i];

    for (int n = 1; n <= N; n++) {
        if (dividend) { /* device_solver.hpp:426-517 */
            /* (quirk) dated in fp64: n * delta_t = 12 * 0.05 = 0.6000000000000001 decides the step */
            const HO_F64 t = n * (HO_F64)delta_t, t_next = (n + 1) * (HO_F64)delta_t;
            const int process = (current_div_idx < p->num_dividends &&
                                 t <= p->div_dates[current_div_idx] &&
                                 p->div_dates[current_div_idx] < t_next);
            if (process) {
                dividend_jump(w, vec_s, U, p->div_amounts[current_div_idx], p->div_percentages[current_div_idx], p->option_type == 1);
                if (p->state_fp32) /* libhadi widens, jumps and rounds the state again */
                    for (int i = 0; i < m; i++) U[i] = (double)(float)U[i];
            }
            if (current_div_idx < p->num_dividends && t > p->div_dates[current_div_idx]) current_div_idx++;
        }
        a0_multiply(w, U, w->A0U);
        a1_multiply(w, U, w->A1U);
        a2_multiply(w, U, w->A2U);
        {
            const double exp_factor = exp(r_f * delta_t * (n - 1));
            if (american)
                for (int i = 0; i < m; i++)
                    w->Y0[i] = U[i] + delta_t * (w->A0U[i] + w->A1U[i] + w->A2U[i] + w->b[i] * exp_factor + lambda_bar[i]);
            else
                for (int i = 0; i < m; i++)
                    w->Y0[i] = U[i] + delta_t * (w->A0U[i] + w->A1U[i] + w->A2U[i] + w->b[i] * exp_factor);
        }
        const double exp_factor_n = exp(r_f * delta_t * n);
        const double exp_factor_nm1 = exp(r_f * delta_t * (n - 1));
        for (int i = 0; i < m; i++)
            w->Y0[i] = w->Y0[i] + theta * delta_t * (w->b1[i] * exp_factor_n - (w->A1U[i] + w->b1[i] * exp_factor_nm1));
        if (dump && dump->step == n) {
            if (dump->A0U) memcpy(dump->A0U, w->A0U, sizeof(double) * m);
            if (dump->A1U) memcpy(dump->A1U, w->A1U, sizeof(double) * m);
            if (dump->A2U) memcpy(dump->A2U, w->A2U, sizeof(double) * m);
            if (dump->Y0rhs) memcpy(dump->Y0rhs, w->Y0, sizeof(double) * m);
        }
        a1_solve(w, w->Y1, w->Y0);
        if (dump && dump->step == n && dump->Y1) memcpy(dump->Y1, w->Y1, sizeof(double) * m);
        for (int i = 0; i < m; i++)
            w->Y1[i] = w->Y1[i] + theta * delta_t * (w->b2[i] * exp_factor_n - (w->A2U[i] + w->b2[i] * exp_factor_nm1));
        if (dump && dump->step == n && dump->Y1rhs) memcpy(dump->Y1rhs, w->Y1, sizeof(double) * m);
        if (f32)
            for (int i = 0; i < m; i++) w->Y1[i] = (double)(float)w->Y1[i];
        a2_solve(w, U, w->Y1);
        if (f32)
            for (int i = 0; i < m; i++) U[i] = (double)(float)U[i];
        if (american) { /* device_solver.hpp:358-372 */
            for (int i = 0; i < m; i++) {
                const double U_bar = U[i];
                U[i] = fmax(U_bar - delta_t * lambda_bar[i], U_0[i]);
                lambda_bar[i] = fmax(0.0, lambda_bar[i] + (U_0[i] - U_bar) / delta_t);
                if (i % (m1 + 1) == m1) lambda_bar[i] = 0.0;
            }
        }
        if (dump && dump->step == n && dump->Unext) memcpy(dump->Unext, U, sizeof(double) * m);
    }
}

static int variant_needs_payoff(int v) { return v == HO_AM || v == HO_AM_DIV; }

int ho_solve(const ho_params *p, const double *vec_s, const double *vec_v,
             const double *delta_s, const double *delta_v,
             double *U, const double *U_0, double *lambda_bar, ho_dump *dump) {
    if (p->m1 < 2 || p->m2 < 3) return -1;
    if (p->scheme == 1 && p->variant != HO_EU) return -4; /* the reference has CS for European only */
    if (p->state_fp32 && ((p->variant != HO_EU && p->variant != HO_DIV) || p->scheme != 0)) return -5;
    double *lam_own = NULL;
    if (variant_needs_payoff(p->variant)) {
        if (!U_0) return -2;
        if (!lambda_bar) { lam_own = zalloc((size_t)(p->m1 + 1) * (p->m2 + 1)); lambda_bar = lam_own; }
    }
    ho_ws *w = ws_new(p->m1, p->m2);
    bc_initialize(w, vec_s, p->r_d, p->r_f, p->N, p->delta_t, p->option_type == 1, p->strike);
    build_all(w, p, vec_s, vec_v, delta_s, delta_v, p->rho, p->sigma, p->kappa, p->eta);
    if (dump) {
        if (dump->b) memcpy(dump->b, w->b, sizeof(double) * w->m);
        if (dump->b1) memcpy(dump->b1, w->b1, sizeof(double) * w->m);
        if (dump->b2) memcpy(dump->b2, w->b2, sizeof(double) * w->m);
    }
    timestepping(w, p, vec_s, U, U_0, lambda_bar, dump);
    ws_free(w);
    free(lam_own);
    return 0;
}

/* The reference's per-operator acceptance drivers, see heston_oracle.h. */
int ho_operator(const ho_params *p, int which,
                const double *vec_s, const double *vec_v, const double *delta_s, const double *delta_v,
                const double *x, const double *b, double *result, double *xsol) {
    if (p->m1 < 2 || p->m2 < 3 || which < 0 || which > 2) return -1;
    ho_ws *w = ws_new(p->m1, p->m2);
    build_all(w, p, vec_s, vec_v, delta_s, delta_v, p->rho, p->sigma, p->kappa, p->eta);
    if (which == 0) {
        if (result) a0_multiply(w, x, result);
    } else if (which == 1) {
        if (result) a1_multiply(w, x, result);
        if (xsol) a1_solve(w, xsol, b);
    } else {
        if (result) a2_multiply(w, x, result);
        if (xsol) a2_solve(w, xsol, b);
    }
    ws_free(w);
    return 0;
}

int ho_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int pick_threads(int threads, int n) {
    int t = threads > 0 ? threads : ho_max_threads();
    if (t > n) t = n;
    if (t < 1) t = 1;
    return t;
}

int ho_solve_batch(const ho_params *p, int n,
                   const double *vec_s, const double *vec_v,
                   const double *delta_s, const double *delta_v,
                   double *U, const double *U_0, double *lambda_bar, int threads) {
    const size_t m = (size_t)(p->m1 + 1) * (p->m2 + 1);
    const int t = pick_threads(threads, n);
#pragma omp parallel for num_threads(t) schedule(dynamic, 1)
    for (int k = 0; k < n; k++) {
        ho_params pk = *p;
        if (p->strike_i) pk.strike = p->strike_i[k];
        ho_solve(&pk, vec_s + (size_t)k * (p->m1 + 1), vec_v + (size_t)k * (p->m2 + 1),
                 delta_s + (size_t)k * p->m1, delta_v + (size_t)k * p->m2,
                 U + k * m, U_0 ? U_0 + k * m : NULL, lambda_bar ? lambda_bar + k * m : NULL, NULL);
    }
    return t;
}

int ho_base_prices(const ho_params *p, int n, double S_0, double V_0, double V, double d,
                   const double *vec_s, double *vec_v,
                   const double *delta_s, double *delta_v,
                   double *U, const double *U_0, double *base_prices, int threads) {
    const int m1 = p->m1, m2 = p->m2;
    const size_t m = (size_t)(m1 + 1) * (m2 + 1);
    const int t = pick_threads(threads, n);
    int bad = 0;
#pragma omp parallel for num_threads(t) schedule(dynamic, 1)
    for (int k = 0; k < n; k++) {
        const double *vs = vec_s + (size_t)k * (m1 + 1);
        double *vv = vec_v + (size_t)k * (m2 + 1), *dvv = delta_v + (size_t)k * m2;
        ho_rebuild_variance(m2, V_0, V, d, vv, dvv);
        ho_params pk = *p;
        if (p->strike_i) pk.strike = p->strike_i[k];
        ho_solve(&pk, vs, vv, delta_s + (size_t)k * m1, dvv, U + k * m, U_0 ? U_0 + k * m : NULL, NULL, NULL);
        const int is = ho_find_s_index(m1, vs, S_0);
        const int iv = ho_find_v_index(m2, vv, V_0);
        if (is < 0) {
#pragma omp atomic write
            bad = 1;
            base_prices[k] = NAN;
        } else {
            base_prices[k] = U[k * m + is + (size_t)iv * (m1 + 1)];
        }
    }
    return bad ? -3 : 0;
}

int ho_jacobian(const ho_params *p, int n, double S_0, double V_0, double V, double d,
                const double *vec_s, double *vec_v,
                const double *delta_s, double *delta_v,
                const double *U_0, double *J, double *base_prices, double eps, int threads) {
    const int m1 = p->m1, m2 = p->m2;
    const size_t m = (size_t)(m1 + 1) * (m2 + 1);
    const int t = pick_threads(threads, n);
    int bad = 0;
#pragma omp parallel for num_threads(t) schedule(dynamic, 1)
    for (int k = 0; k < n; k++) {
        const double *vs = vec_s + (size_t)k * (m1 + 1), *dss = delta_s + (size_t)k * m1;
        double *vv = vec_v + (size_t)k * (m2 + 1), *dvv = delta_v + (size_t)k * m2;
        const double *u0 = U_0 + k * m;
        double *U = zalloc(m), *lam = zalloc(m);
        ho_ws *w = ws_new(m1, m2);
        ho_rebuild_variance(m2, V_0, V, d, vv, dvv);
        ho_params pkv = *p;
        if (p->strike_i) pkv.strike = p->strike_i[k];
        const ho_params *const pk = &pkv;
        bc_initialize(w, vs, pk->r_d, pk->r_f, pk->N, pk->delta_t, pk->option_type == 1, pk->strike); /* built once (jacobian_computation.cpp:255) */
        const int is = ho_find_s_index(m1, vs, S_0);
        const int iv = ho_find_v_index(m2, vv, V_0);
        if (is < 0) {
#pragma omp atomic write
            bad = 1;
        }
        const int isx = is < 0 ? 0 : is;
        memcpy(U, u0, sizeof(double) * m);
        build_all(w, pk, vs, vv, dss, dvv, p->rho, p->sigma, p->kappa, p->eta);
        timestepping(w, pk, vs, U, u0, lam, NULL);
        const double base = U[isx + (size_t)iv * (m1 + 1)];
        base_prices[k] = base;
        for (int param = 0; param < 4; param++) {
            double kappa_p = p->kappa, eta_p = p->eta, sigma_p = p->sigma, rho_p = p->rho;
            switch (param) {
                case 0: kappa_p += eps; break;
                case 1: eta_p += eps; break;
                case 2: sigma_p += eps; break;
                case 3: rho_p += eps; break;
            }
            memcpy(U, u0, sizeof(double) * m);
            build_all(w, pk, vs, vv, dss, dvv, rho_p, sigma_p, kappa_p, eta_p);
            timestepping(w, pk, vs, U, u0, lam, NULL);
            J[(size_t)k * 5 + param] = (U[isx + (size_t)iv * (m1 + 1)] - base) / eps;
        }
        memcpy(U, u0, sizeof(double) * m);
        ho_rebuild_variance(m2, V_0 + eps, V, d, vv, dvv); /* boundary vectors NOT rebuilt */
        const int ivp = ho_find_v_index(m2, vv, V_0 + eps);
        build_all(w, pk, vs, vv, dss, dvv, p->rho, p->sigma, p->kappa, p->eta);
        timestepping(w, pk, vs, U, u0, lam, NULL);
        J[(size_t)k * 5 + 4] = (U[isx + (size_t)ivp * (m1 + 1)] - base) / eps;
        ws_free(w);
        free(U); free(lam);
    }
    return bad ? -3 : 0;
}

/* ---- jacobian_computation.cpp:20-195 ---------------------------------------- */
void ho_lm_update(int n, const double *J, const double *residuals, double lambda, double *delta) {
    enum { NP = 5 };
    double A[NP * NP], b[NP];
    for (int i = 0; i < NP; i++)
        for (int j = 0; j < NP; j++) {
            double s = 0.0;
            for (int k = 0; k < n; k++) s += J[(size_t)k * NP + i] * J[(size_t)k * NP + j];
            A[i * NP + j] = s;
        }
    for (int i = 0; i < NP; i++) A[i * NP + i] *= (1.0 + lambda);
    for (int i = 0; i < NP; i++) {
        double s = 0.0;
        for (int k = 0; k < n; k++) s += J[(size_t)k * NP + i] * residuals[k];
        b[i] = s;
    }
    for (int k = 0; k < NP; k++) {
        double maxA = fabs(A[k * NP + k]);
        int pivotRow = k;
        for (int r = k + 1; r < NP; r++) {
            const double val = fabs(A[r * NP + k]);
            if (val > maxA) { maxA = val; pivotRow = r; }
        }
        if (pivotRow != k) {
            for (int col = 0; col < NP; col++) {
                const double tmp = A[k * NP + col];
                A[k * NP + col] = A[pivotRow * NP + col];
                A[pivotRow * NP + col] = tmp;
            }
            const double tmpb = b[k]; b[k] = b[pivotRow]; b[pivotRow] = tmpb;
        }
        const double pivot = A[k * NP + k];
        for (int col = k + 1; col < NP; col++) A[k * NP + col] /= pivot;
        b[k] /= pivot;
        A[k * NP + k] = 1.0;
        for (int i = k + 1; i < NP; i++) {
            const double factor = A[i * NP + k];
            for (int col = k + 1; col < NP; col++) A[i * NP + col] -= factor * A[k * NP + col];
            b[i] -= factor * b[k];
            A[i * NP + k] = 0.0;
        }
    }
    for (int k = NP - 1; k >= 0; k--) {
        double val = b[k];
        for (int col = k + 1; col < NP; col++) val -= A[k * NP + col] * b[col];
        b[k] = val;
    }
    for (int i = 0; i < NP; i++) delta[i] = b[i];
}
