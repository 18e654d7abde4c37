// hadi_core.h -- internal HBM layout and per-instance operator tables of libhadi.
//
// Everything the reference stores per grid point (A0 9 values/pt, A1 7 arrays, A2 13 arrays,
// 4 dense boundary vectors: hes_a0_kernels.hpp:22, hes_a1_kernels.hpp:21-30,
// hes_a2_shuffled_kernels.hpp:58-75, hes_boundary_kernels.hpp:11-14) is kept here in
// O(m1 + m2) numbers per instance, because
//   A0(i,j;k,l) = rho*sigma * [s_i beta_s(i-1,k)] * [v_j beta_v(j-1,l)]      (separable)
//   A1(i,j;k)   = v_j * [1/2 s_i^2 delta_s(i-1,k)] + (r_d-r_f) * [s_i beta_s(i-1,k)] (- 1/2 r_d)
//   A2(j;.)     does not depend on i
//   b1, b2      are sparse (one entry per v-row / the last v-row), b0 == 0.
// The functions are HADI_HD so tests can check the tables on the CPU against the oracle; the
// product only ever runs them inside hadi_setup_kernel on the GPU.
#pragma once
#include "hadi_device.h"

#define HADI_RC 16   // doubles per v-row in the row table
#define HADI_RCL 12  // of which the first 12 are used (compact stride of the LDS copies)
#define HADI_PBW 12  // doubles per v-row in the column-pass table
#define HADI_MAX_P 16 // max chunks (waves) per column in the column pass
#define HADI_LC 33   // max rows per chunk in the column pass
// Zero pad slots behind the i = 0 slot of every row (the right neighbour of i = m1 reads the first one).  From 4 nodes per
// lane on, the pitch is padded so that every v-row of the STATE arrays starts on a 128-byte line -- a multiple of 16
// doubles, or of 32 floats for the fp32 state: the column pass reads 64-column row segments, which then cover whole lines
// instead of straddling one more on every other row (measured per launch: 512x256 x256 column pass 0.1137 -> 0.1093 ms,
// 1024x512 x64: 0.163 -> 0.143 ms, with the fp32 state 0.102 -> 0.082 ms; HBM reads of the pass 9.3 -> 8.7 B per point).
// ES = bytes per element of the state arrays (8, or 4 for HADI_STATE_FP32).
#define HADI_ROW_PAD(B, ES) ((B) < 4 ? 8 : ((ES) == 4 ? 32 : 16))
#define HADI_B1_BOTH (1 << 24)  // row-table flag added to RC_B1COL (a column index < 2^20): the row has a b1 entry at column 0 as well (m2 > m1 only)

// Column pass with the reduced system on the matrix core (hadi_pb_solve<.., MF>): LDS doubles = exchange values Z [4 P][64] +
// selected inverse rows, transposed, RT [4 P][MP] + the product T [MP][64], MP = 4 P rounded up to whole 16-row blocks
HADI_HD inline int hadi_pb_mp(int P) { return 16 * ((P + 3) >> 2); }
HADI_HD inline size_t hadi_pb_mf_doubles(int P) { return (size_t)4 * P * 64 + (size_t)4 * P * hadi_pb_mp(P) + (size_t)hadi_pb_mp(P) * 64; }
#ifndef HADI_PB_MF
#define HADI_PB_MF 1
#endif

struct HadiLayout {
    int m1, m2, nrows;  // nrows = m2 + 1
    int nrows_pad;      // P * HADI_LC >= nrows: v-rows past nrows are identity rows (always 0) so that
                        // every column-pass wavefront owns exactly HADI_LC rows -- no tail branches
    int B;              // grid points per lane in the row pass
    int G;              // wavefronts per v-row in the row pass: wave g owns i = 1 + 64*B*g + B*lane + r
    int rowp;           // row pitch in doubles: 64*B*G + 8 (slot 64*B*G holds i = 0, 7 zero pads)
    int P;              // chunks per column in the column pass
    int lc;             // rows per chunk: HADI_LC, or nrows (one chunk: P = 1) for the sequential column pass of grids with
                        // more than HADI_MAX_P * HADI_LC v-rows (hadi_pass_b_seq)
    long long inst_stride;  // rowp * nrows_pad
};

// Storage position of s-index i inside a row.  Lane l of wave g of the row pass owns the B nodes
// i = 1 + 64*B*g + B*l .. and fetches them as B/2 coalesced 16-byte accesses: pair q of (g, l) sits at
// q*128*G + 128*g + 2*l (one wave-instruction = 1 KiB contiguous).
HADI_HD inline int hadi_pos(int B, int G, int i) {
    if (i == 0) return 64 * B * G;
    const int e = i - 1;
    const int g = e / (64 * B), el = e - g * 64 * B;
    if (B == 1) return 64 * g + el;
    const int lane = el / B, r = el % B;
    return (r >> 1) * 128 * G + 128 * g + 2 * lane + (r & 1);
}
HADI_HD inline int hadi_pos(const HadiLayout &L, int i) { return hadi_pos(L.B, L.G, i); }
// Slot mapping of the fp32 state arrays: same idea with QUADS of nodes, so that a lane's accesses stay 16 bytes wide
// (4 floats).  B >= 4: quad q of (wave g, lane l) sits at q*256*G + 256*g + 4*l; B < 4: as hadi_pos.
HADI_HD inline int hadi_pos_f32(int B, int G, int i) {
    if (B < 4 || i == 0) return hadi_pos(B, G, i);
    const int e = i - 1;
    const int g = e / (64 * B), el = e - g * 64 * B;
    const int lane = el / B, r = el % B;
    return (r >> 2) * 256 * G + 256 * g + 4 * lane + (r & 3);
}

// Inverse: s-index stored in `slot` (-1 for a pad slot).
HADI_HD inline int hadi_slot_to_i(const HadiLayout &L, int slot) {
    const int B = L.B, G = L.G, n = 64 * B * G;
    if (slot == n) return 0;
    if (slot > n) return -1;
    if (B == 1) return slot + 1;  // slot = 64 g + lane = e
    const int q = slot / (128 * G), rem = slot - q * 128 * G;
    const int g = rem >> 7, rem2 = rem & 127, lane = rem2 >> 1, r = 2 * q + (rem2 & 1);
    return 1 + 64 * B * g + B * lane + r;
}

// Row-pass shape for m1 s-intervals.  One wavefront per row up to 512 nodes (8 per lane); measured on
// MI355X at m1 = 512: (B, G) = (8, 1) at 2 waves/SIMD runs 0.183 ms/launch, (4, 2) at 4 waves/SIMD 0.249 ms
// (the split solve costs 1.75x the instructions).  Two wavefronts per row above 512 nodes.  Above 1024 nodes the row is kept
// in natural order (B = 1: slot of node i >= 1 is i - 1) with as many 64-slot groups as it needs, for the sequential row
// pass (hadi_pass_a_seq: lane <-> v-row).
HADI_HD inline void hadi_pick_shape(int m1, int *B, int *G) {
    if (m1 <= 64) { *B = 1; *G = 1; }
    else if (m1 <= 128) { *B = 2; *G = 1; }
    else if (m1 <= 256) { *B = 4; *G = 1; }
    else if (m1 <= 512) { *B = 8; *G = 1; }
    else if (m1 <= 1024) { *B = 8; *G = 2; }
    else { *B = 1; *G = (m1 + 63) / 64; }
}

struct HadiInstPar {
    double dt, thdt;   // delta_t, theta*delta_t
    double q;          // r_d - r_f
    double half_rd;    // 0.5*r_d
    double bc_rate;    // boundary data carry the time factor e_n = exp(bc_rate dt n): r_f for the call
                       // (device_solver.hpp:238,246), -r_d for the put (u = K e^{-r_d t} on the Dirichlet edges)
    double hr0;        // reaction term of the i = 0 row of A1: 0 for the call (hes_a1_kernels.hpp:56-61 leaves the row
                       // empty), 0.5*r_d for the put (see hadi_option_type in hadi.h)
    int N;             // time steps of this instance
    int idx_s, idx_v;  // price node (filled by the pick step), -1 if S_0 is off-grid
    int put;           // 1 = put boundary data (hadi.h, enum hadi_option_type)
};

// ---- coeff.hpp:24-126 ------------------------------------------------------------------------
HADI_HD inline double hadi_fd_delta(const double *D, int i, int pos) {
    if (pos == -1) return 2 / (D[i] * (D[i] + D[i + 1]));
    if (pos == 0) return -2 / (D[i] * D[i + 1]);
    return 2 / (D[i + 1] * (D[i] + D[i + 1]));
}
HADI_HD inline double hadi_fd_beta(const double *D, int i, int pos) {
    if (pos == -1) return -D[i + 1] / (D[i] * (D[i] + D[i + 1]));
    if (pos == 0) return (D[i + 1] - D[i]) / (D[i] * D[i + 1]);
    return D[i] / (D[i + 1] * (D[i] + D[i + 1]));
}
HADI_HD inline double hadi_fd_alpha(const double *D, int i, int pos) {
    if (pos == -2) return D[i] / (D[i - 1] * (D[i - 1] + D[i]));
    if (pos == -1) return (-D[i - 1] - D[i]) / (D[i - 1] * D[i]);
    return (D[i - 1] + 2 * D[i]) / (D[i] * (D[i - 1] + D[i]));
}
HADI_HD inline double hadi_fd_gamma(const double *D, int i, int pos) {
    if (pos == 0) return (-2 * D[i + 1] - D[i + 2]) / (D[i + 1] * (D[i + 1] + D[i + 2]));
    if (pos == 1) return (D[i + 1] + D[i + 2]) / (D[i + 1] * D[i + 2]);
    return -D[i + 1] / (D[i + 2] * (D[i + 1] + D[i + 2]));
}

// Row r of the explicit A2 operator, accumulated in the reference's order
// (hes_a2_shuffled_kernels.hpp:122-152): the upwind block of iteration j = r-1 lands on row r
// BEFORE iteration r adds the reaction term and its own stencil.  out = {lo2, lo, mn, up, up2}.
HADI_HD inline void hadi_a2_row(int r, int m2, const double *vv, const double *dv, double r_d,
                                double kappa, double eta, double sigma, double *out) {
    double lo2 = 0.0, lo = 0.0, mn = 0.0, up = 0.0, up2 = 0.0;
    const int j = r - 1;
    if (j >= 1 && j < m2 - 1 && vv[j] > 1.0) {
        const double temp = kappa * (eta - vv[j]);
        const double temp2 = 0.5 * sigma * sigma * vv[j];
        lo2 += temp * hadi_fd_alpha(dv, j, -2);
        lo += temp * hadi_fd_alpha(dv, j, -1);
        mn += temp * hadi_fd_alpha(dv, j, 0);
        lo += temp2 * hadi_fd_delta(dv, j - 1, -1);
        mn += temp2 * hadi_fd_delta(dv, j - 1, 0);
        up += temp2 * hadi_fd_delta(dv, j - 1, 1);
    }
    if (r < m2 - 1) {
        const double temp = kappa * (eta - vv[r]);
        const double temp2 = 0.5 * sigma * sigma * vv[r];
        mn += -0.5 * r_d;
        if (r == 0) {
            mn += temp * hadi_fd_gamma(dv, 0, 0);
            up += temp * hadi_fd_gamma(dv, 0, 1);
            up2 += temp * hadi_fd_gamma(dv, 0, 2);
        } else {
            lo += temp * hadi_fd_beta(dv, r - 1, -1) + temp2 * hadi_fd_delta(dv, r - 1, -1);
            mn += temp * hadi_fd_beta(dv, r - 1, 0) + temp2 * hadi_fd_delta(dv, r - 1, 0);
            up += temp * hadi_fd_beta(dv, r - 1, 1) + temp2 * hadi_fd_delta(dv, r - 1, 1);
        }
    }
    out[0] = lo2; out[1] = lo; out[2] = mn; out[3] = up; out[4] = up2;
}

// Inputs of one instance's setup.
struct HadiSetupIn {
    const double *vec_s, *vec_v, *delta_s, *delta_v;  // this instance's grid
    double r_d, r_f, rho, sigma, kappa, eta, theta, dt;
    int N;
    int put;        // 0 = call boundary data (the reference), 1 = put (hadi.h, enum hadi_option_type)
    double strike;  // put only
};

// Output tables of one instance (all device pointers, already offset to the instance).
struct HadiTables {
    double *scoef;   // [4][64*B*G] Bm, Bp, Dm, Dp with B_k = s_i beta_s(i-1,k), D_k = 1/2 s_i^2 delta_s(i-1,k),
                     //             k = -1, +1 (the k = 0 weights are -(m + p): the FD weights sum to zero)
    double *b2row;   // [rowp]      -1/2 r_d s_i E in state layout (hes_boundary_kernels.hpp:62-66)
    double *rowc;    // [nrows][HADI_RC]
    double *a2i;     // [5][nrows_pad]  implicit A2 diagonals by row: l2, l1, m, u1, u2 (identity past nrows)
    double *pb;      // [nrows_pad][HADI_PBW]  column-pass factorisation + spikes
    double *rinv;    // [4P][4P]    inverse of the SPIKE reduced matrix (P > 1)
    double *rwork;   // [4P][8P]    Gauss-Jordan work area
    HadiInstPar *ipar;
};

// rowc columns
// wavefronts (= strips) per block of the strip row pass: 8 at 8 nodes per lane (one block fills a CU's LDS and
// registers); 4 at 4 and 2 nodes per lane, where a v-line is short and 8 strips would be ~17 rows each
#ifndef HADI_STRIP_WAVES_B4
#define HADI_STRIP_WAVES_B4 4
#endif
#define HADI_STRIP_WAVES(B) ((B) == 8 ? 8 : (B) == 4 ? HADI_STRIP_WAVES_B4 : 4)
// Slots of a strip wavefront's private LDS ring = rows ahead of the current one that are fetched or in flight (the row NS
// ahead is issued when the row two ahead is waited for: a lead of NS - 2 row steps).  8 nodes per lane: 4 slots fill the
// 160 KB of a CU (3 for the paired strips with an fp64 state).  Narrower rows have LDS to spare, but a deeper ring does not
// pay: measured with 8 slots at 4 nodes per lane, 256x128 x512 American puts 0.0841 -> 0.0871 ms per launch, 200x100 x700
// 0.076 -> 0.116 (the LDS costs an occupancy step) -- the DMA wait the in-kernel stamps show is not what bounds the pass.
#ifndef HADI_STRIP_NS_NARROW
#define HADI_STRIP_NS_NARROW 4
#endif
#define HADI_STRIP_NS(B, G, ES) ((G) == 2 ? ((ES) == 8 ? 3 : 4) : ((B) <= 4 ? HADI_STRIP_NS_NARROW : 4))
enum { RC_V = 0, RC_WM = 1, RC_WZ = 2, RC_WP = 3, RC_L2 = 4, RC_L1 = 5, RC_M = 6, RC_U1 = 7, RC_U2 = 8,
       RC_B1VAL = 9, RC_B1COL = 10, RC_VTH = 11 /* theta dt v */, RC_LAST = 12,
       // the A0 v-weights divided by -theta dt (r_d - r_f): the strip kernels keep -theta dt (r_d - r_f) s beta_s in place of
       // s beta_s (hadi_strip_step) and load the 12 entries RC_L2 .. RC_WPS of a row
       RC_WMS = 13, RC_WZS = 14, RC_WPS = 15 };
#define HADI_SRC0 RC_L2  // first entry of the strip kernels' 12-entry window of a row-table row
// pb columns: forward  y_k = (rhs_k - PB_L y_{k-1} - PB_L2 y_{k-2}) * PB_Q
//             backward x_k = y_k - PB_C x_{k+1} - PB_C2 x_{k+2}
//             spikes   x_k -= PB_V0 tl0 + PB_V1 tl1 + PB_W0 tr0 + PB_W1 tr1
// The column kernels run the forward sweep as y_k = fma(-PB_LQ, y_{k-1}, fma(-PB_L2Q, y_{k-2}, rhs_k PB_Q)) with the scaled
// multipliers PB_LQ = PB_L PB_Q, PB_L2Q = PB_L2 PB_Q: ONE operation of the step depends on the step before instead of three
// (and the backward sweep as fma(-PB_C, x_{k+1}, fma(-PB_C2, x_{k+2}, y_k)): one instead of two).
enum { PB_L = 0, PB_L2 = 1, PB_Q = 2, PB_C = 3, PB_C2 = 4, PB_V0 = 5, PB_V1 = 6, PB_W0 = 7, PB_W1 = 8, PB_LQ = 9, PB_L2Q = 10 };

struct HadiNoSync { HADI_HD void operator()() const {} };

// Builds every table of one instance.  Work is split over `nth` cooperating threads
// (tid = 0..nth-1) with `sync()` between phases; nth = 1 with HadiNoSync runs it serially.
template <class Sync>
HADI_HD inline void hadi_setup_instance(const HadiLayout &L, const HadiSetupIn &in, const HadiTables &t,
                                        int tid, int nth, Sync sync) {
    const int m1 = L.m1, m2 = L.m2, nrows = L.nrows, npad = L.nrows_pad, nslot = 64 * L.B * L.G;
    // call: hes_boundary_kernels.hpp:56.  put: the boundary value K e^{-r_d dt n} carries its whole time factor in e_n
    const double E = in.put ? 1.0 : exp(-in.r_f * in.dt * (in.N - 1));
    const double thdt = in.theta * in.dt;

    if (tid == 0) {
        HadiInstPar ip;
        ip.dt = in.dt; ip.thdt = thdt; ip.q = in.r_d - in.r_f; ip.half_rd = 0.5 * in.r_d;
        ip.bc_rate = in.put ? -in.r_d : in.r_f;
        ip.hr0 = in.put ? 0.5 * in.r_d : 0.0;
        ip.N = in.N; ip.idx_s = -1; ip.idx_v = 0; ip.put = in.put;
        *t.ipar = ip;
    }
    // --- s-direction coefficients (hes_a0_kernels.hpp:37-49, hes_a1_kernels.hpp:69-91) ---------
    for (int k = tid; k < 4 * nslot; k += nth) t.scoef[k] = 0.0;
    for (int k = tid; k < L.rowp; k += nth) t.b2row[k] = 0.0;
    sync();
    for (int i = tid + 1; i < m1; i += nth) {
        const double s = in.vec_s[i];
        const int pos = hadi_pos(L, i);
        t.scoef[0 * nslot + pos] = s * hadi_fd_beta(in.delta_s, i - 1, -1);
        t.scoef[1 * nslot + pos] = s * hadi_fd_beta(in.delta_s, i - 1, 1);
        t.scoef[2 * nslot + pos] = 0.5 * s * s * hadi_fd_delta(in.delta_s, i - 1, -1);
        t.scoef[3 * nslot + pos] = 0.5 * s * s * hadi_fd_delta(in.delta_s, i - 1, 1);
    }
    // b2 on the last v-row: the value that makes the far-field solution exact there -- u = s e^{-r_f t} for the call
    // (hes_boundary_kernels.hpp:62-66), u = K e^{-r_d t} for the put
    for (int i = tid; i <= m1; i += nth) t.b2row[hadi_pos(L, i)] = -0.5 * in.r_d * (in.put ? in.strike : in.vec_s[i]) * E;
    // --- v-rows: A0 weights, explicit A2, boundary b1 ---------------------------------------------
    for (int r = tid; r < nrows; r += nth) {
        double *rc = t.rowc + (size_t)r * HADI_RC;
        for (int k = 0; k < HADI_RC; k++) rc[k] = 0.0;
        const double v = in.vec_v[r];
        rc[RC_V] = v;
        rc[RC_VTH] = thdt * v;
        if (r >= 1 && r <= m2 - 1) {
            const double c = in.rho * in.sigma * v;
            rc[RC_WM] = c * hadi_fd_beta(in.delta_v, r - 1, -1);
            rc[RC_WZ] = c * hadi_fd_beta(in.delta_v, r - 1, 0);
            rc[RC_WP] = c * hadi_fd_beta(in.delta_v, r - 1, 1);
            const double qth = thdt * (in.r_d - in.r_f);
            if (qth != 0.0) {  // (r_d == r_f: the host keeps such batches off the strip kernels)
                rc[RC_WMS] = -rc[RC_WM] / qth;
                rc[RC_WZS] = -rc[RC_WZ] / qth;
                rc[RC_WPS] = -rc[RC_WP] / qth;
            }
        }
        double a2[5];
        hadi_a2_row(r, m2, in.vec_v, in.delta_v, in.r_d, in.kappa, in.eta, in.sigma, a2);
        for (int k = 0; k < 5; k++) rc[RC_L2 + k] = a2[k];
        rc[RC_B1COL] = -1.0;
        rc[RC_LAST] = (r == m2) ? 1.0 : 0.0;
        // implicit diagonals I - theta*dt*A2 (hes_a2_shuffled_kernels.hpp:159-171)
        t.a2i[0 * npad + r] = -in.theta * in.dt * a2[0];
        t.a2i[1 * npad + r] = -in.theta * in.dt * a2[1];
        t.a2i[2 * npad + r] = 1.0 - in.theta * in.dt * a2[2];
        t.a2i[3 * npad + r] = -in.theta * in.dt * a2[3];
        t.a2i[4 * npad + r] = -in.theta * in.dt * a2[4];
    }
    for (int r = nrows + tid; r < npad; r += nth) {  // padding rows: identity, decoupled
        t.a2i[0 * npad + r] = 0.0; t.a2i[1 * npad + r] = 0.0; t.a2i[2 * npad + r] = 1.0;
        t.a2i[3 * npad + r] = 0.0; t.a2i[4 * npad + r] = 0.0;
    }
    sync();
    // b1_(m1*(j+1)), j = 0..m2 (quirk: not idx(m1,j)), hes_boundary_kernels.hpp:54-58: every entry has the same value.
    // v-row r holds the multiples of m1 inside [r(m1+1), r(m1+1) + m1]: one for r < m1 (column m1 - r), and -- only
    // when m2 > m1 -- two on the rows r = k m1 (columns 0 AND m1), one on the rows in between.  RC_B1COL = the column
    // (-1: none), + HADI_B1_BOTH when column 0 carries a second entry.
    for (int r = tid; r < nrows; r += nth) {
        double *rc = t.rowc + (size_t)r * HADI_RC;
        const long long lo = (long long)r * (m1 + 1), hi = lo + m1;
        long long q = (lo + m1 - 1) / m1;  // first multiple of m1 >= lo is q*m1
        if (q < 1) q = 1;                  // j + 1 >= 1
        int ncol = 0, cols[2] = {-1, -1};
        for (; q * m1 <= hi && q <= (long long)m2 + 1 && ncol < 2; q++) cols[ncol++] = (int)(q * m1 - lo);
        if (ncol > 0) rc[RC_B1VAL] = in.put ? 0.0 : (in.r_d - in.r_f) * in.vec_s[m1] * E;  // put: du/ds = 0 at s_max
        if (ncol == 1) rc[RC_B1COL] = (double)cols[0];
        if (ncol == 2) rc[RC_B1COL] = (double)(cols[1] + HADI_B1_BOTH);  // cols = {0, m1}
    }
    // --- column pass: per-chunk pentadiagonal LU + SPIKE vectors ----------------------------------
    const double *l2 = t.a2i, *l1 = t.a2i + npad, *dm = t.a2i + 2 * npad, *u1 = t.a2i + 3 * npad,
                 *u2 = t.a2i + 4 * npad;
    const int P = L.P;
    for (int p = tid; p < P; p += nth) {
        const int ja = p * L.lc, len = L.lc;
        double c1 = 0.0, c21 = 0.0;  // c, c2 of row k-1
        double c0 = 0.0, c20 = 0.0;  // c, c2 of row k-2
        for (int k = 0; k < len; k++) {
            const int j = ja + k;
            double *pb = t.pb + (size_t)j * HADI_PBW;
            const double e2 = (k >= 2) ? l2[j] : 0.0;          // coupling to k-2 inside the chunk
            const double Lk = (k >= 1) ? (l1[j] - e2 * c0) : 0.0;
            const double den = dm[j] - Lk * c1 - e2 * c20;
            const double q = 1.0 / den;
            const double uu1 = (k + 1 < len) ? u1[j] : 0.0;
            const double uu2 = (k + 2 < len) ? u2[j] : 0.0;
            const double c = (uu1 - Lk * c21) * q;
            const double c2 = uu2 * q;
            pb[PB_L] = Lk; pb[PB_L2] = e2; pb[PB_Q] = q; pb[PB_C] = c; pb[PB_C2] = c2;
            for (int z = PB_V0; z < HADI_PBW; z++) pb[z] = 0.0;
            pb[PB_LQ] = Lk * q; pb[PB_L2Q] = e2 * q;
            c0 = c1; c20 = c21; c1 = c; c21 = c2;
        }
    }
    sync();
    // spikes: thread (p, w) solves the chunk system for one coupling column.
    //   V = M_pp^-1 [coupling to the previous chunk's last two unknowns x[ja-2], x[ja-1]]
    //   W = M_pp^-1 [coupling to the next chunk's first two unknowns x[jb], x[jb+1]]
    for (int pw = tid; pw < 4 * P; pw += nth) {
        const int p = pw >> 2, w = pw & 3;
        const int ja = p * L.lc, len = L.lc;
        if ((w < 2 && p == 0) || (w >= 2 && p == P - 1)) continue;
        const int jb = ja + len;
        // forward sweep with rhs given on the fly
        double y1 = 0.0, y0 = 0.0;  // y_{k-1}, y_{k-2}
        for (int k = 0; k < len; k++) {
            const int j = ja + k;
            double rhs = 0.0;
            if (w == 0 && k == 0) rhs = l2[ja];
            if (w == 1 && k == 0) rhs = l1[ja];
            if (w == 1 && k == 1) rhs = l2[ja + 1];
            if (w == 2 && k == len - 2) rhs = u2[jb - 2];
            if (w == 2 && k == len - 1) rhs = u1[jb - 1];
            if (w == 3 && k == len - 1) rhs = u2[jb - 1];
            double *pb = t.pb + (size_t)j * HADI_PBW;
            const double y = (rhs - pb[PB_L] * y1 - pb[PB_L2] * y0) * pb[PB_Q];
            pb[PB_V0 + w] = y;
            y0 = y1; y1 = y;
        }
        double x1 = 0.0, x2 = 0.0;  // x_{k+1}, x_{k+2}
        for (int k = len - 1; k >= 0; k--) {
            double *pb = t.pb + (size_t)(ja + k) * HADI_PBW;
            const double x = pb[PB_V0 + w] - pb[PB_C] * x1 - pb[PB_C2] * x2;
            pb[PB_V0 + w] = x;
            x2 = x1; x1 = x;
        }
    }
    sync();
    if (P > 1) {
        // Reduced system R t = z on the 4P interface unknowns {first2, last2} of every chunk;
        // invert it once by Gauss-Jordan (R = I + small couplings, cond ~ 3: no pivoting needed).
        const int n4 = 4 * P, ldw = 2 * n4;
        double *Wk = t.rwork;
        for (int e = tid; e < n4 * ldw; e += nth) {
            const int r = e / ldw, c = e % ldw;
            Wk[e] = (c == r || c == n4 + r) ? 1.0 : 0.0;
        }
        sync();
        for (int pq = tid; pq < n4; pq += nth) {
            const int p = pq >> 2, qd = pq & 3;
            const int ja = p * L.lc, len = L.lc;
            const int lrow = (qd == 0) ? 0 : (qd == 1) ? 1 : (qd == 2) ? len - 2 : len - 1;
            const double *pb = t.pb + (size_t)(ja + lrow) * HADI_PBW;
            if (p > 0) {
                Wk[pq * ldw + 4 * (p - 1) + 2] += pb[PB_V0];
                Wk[pq * ldw + 4 * (p - 1) + 3] += pb[PB_V1];
            }
            if (p < P - 1) {
                Wk[pq * ldw + 4 * (p + 1) + 0] += pb[PB_W0];
                Wk[pq * ldw + 4 * (p + 1) + 1] += pb[PB_W1];
            }
        }
        sync();
        for (int k = 0; k < n4; k++) {
            const double piv = 1.0 / Wk[k * ldw + k];
            sync();
            for (int c = tid; c < ldw; c += nth) Wk[k * ldw + c] *= piv;
            sync();
            for (int e = tid; e < n4 * ldw; e += nth) {
                const int r = e / ldw, c = e % ldw;
                if (r != k && c != k) Wk[e] -= Wk[r * ldw + k] * Wk[k * ldw + c];
            }
            sync();
            for (int r = tid; r < n4; r += nth)
                if (r != k) Wk[r * ldw + k] = 0.0;
            sync();
        }
        for (int e = tid; e < n4 * n4; e += nth) t.rinv[e] = Wk[(e / n4) * ldw + n4 + (e % n4)];
    }
    sync();
}
