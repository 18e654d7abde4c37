// hadi_k_row_ring.h -- row pass on a shared LDS ring (hadi_row_step, hadi_pass_a): every shape up to m1 = 1024, Craig-Sneyd predictor / corrector, the row step of the LDS-resident block kernel.
// Part of libhadi's device code: include through hadi_kernels.h (which fixes the order).
#pragma once

// pass A.  Block = W*G wavefronts working on W consecutive v-rows of one instance at a time, G wavefronts
// per row.  The rows j-2 .. j+W+1 they need (9-point A0, 5-point A2) sit in an LDS ring of 2W+4 rows that is
// refilled by LDS-DMA one iteration ahead, so HBM latency hides behind the line solves.  Lane l of wave g owns
// the s-nodes i = 1 + 64*B*g + B*l .. of its row.  The s-direction coefficient arrays live in LDS too.
// With G = 2 the row's tridiagonal system is split at the wave boundary: each half is solved against one
// unknown boundary value (a second right-hand side carried through the cyclic reduction) and the two
// boundary values follow from a 2x2 system exchanged through LDS -- small per-lane state (B = 4 at
// m1 = 512) is what lets four wavefronts share a SIMD.
template <class T>
struct HadiRowCtxT {
    const double *coef;  // LDS: Bm, Bp, Dm, Dp, each 64*B*G doubles in row layout
    double *xch;         // LDS: [W][8] boundary exchange between the two waves of a row (G = 2): 4 values + 2 tokens
    T *Yi;               // instance base of Y (T = float: fp32-state sweep)
    const double *Li;    // instance base of lambda_bar (American)
    const double *rowc;  // LDS copy of the row table of this block's tile: entry (j - j0)
    int j0;              // first v-row of the tile
    const double *b2r;   // instance b2 row (global)
    double *R1i, *C2i;   // instance bases of the Craig-Sneyd carry-over arrays (MODE 1 writes, MODE 2 reads)
    int lane, half, wrow, posL, posR, rowp;
    double dt, thdt, qd, half_rd, e_nm1, e_n;
    double hr0, inv0;    // i = 0 row of A1: reaction term (0 for the call) and 1 / (1 + theta dt hr0)
    // American without the lambda_bar array (AMER == 2, see hadi_row_step): LDS copy of the payoff row (it depends on s
    // only), 1/dt, and which (lane, node) of this wavefront is i = m1 (lambda_bar is forced to 0 there), -1 if none
    const double *payrow;
    double inv_dt;
    int m1_lane, m1_r;
    int *err;            // HadiSweepArgs.err / .debug (used by the G = 2 rendezvous only)
    int debug;
    HADI_STAMP_ACC
};
typedef HadiRowCtxT<double> HadiRowCtx;

// Loads this lane's B values of a row-layout array (LDS or global): pair q at q*128*G + 128*half + 2*lane.
// T = double (16-byte pairs) or float (fp32 state: 8-byte pairs, widened on load / rounded on store).
template <class T> struct HadiPair;
template <> struct HadiPair<double> { typedef double2 type; };
template <> struct HadiPair<float> { typedef float2 type; };
template <int B, int G, class T = double>
HADI_DEV HADI_FORCEINLINE void hadi_get_block(const T *row, int half, int lane, double (&u)[B]) {
    if constexpr (sizeof(T) == 4 && B >= 4) {  // fp32 state: quads (hadi_pos_f32), 16-byte accesses
#pragma unroll
        for (int q = 0; q < B / 4; q++) {
            const float4 t = *reinterpret_cast<const float4 *>(row + q * 256 * G + 256 * half + 4 * lane);
            u[4 * q] = (double)t.x; u[4 * q + 1] = (double)t.y; u[4 * q + 2] = (double)t.z; u[4 * q + 3] = (double)t.w;
        }
    } else if constexpr (B == 1) {
        u[0] = (double)row[64 * half + lane];
    } else {
#pragma unroll
        for (int q = 0; q < B / 2; q++) {
            const typename HadiPair<T>::type t =
                *reinterpret_cast<const typename HadiPair<T>::type *>(row + q * 128 * G + 128 * half + 2 * lane);
            u[2 * q] = (double)t.x;
            u[2 * q + 1] = (double)t.y;
        }
    }
}
#ifndef HADI_ROW_ST_NT   // (experiment: the row pass's result stores non-temporal too -- they are meant to STAY in the memory-side cache)
#define HADI_ROW_ST_NT 0
#endif
template <int B, int G, class T = double>
HADI_DEV HADI_FORCEINLINE void hadi_put_block(T *row, int half, int lane, const double (&u)[B]) {
#if HADI_ROW_ST_NT && !defined(HADI_EMU)
    if constexpr (sizeof(T) == 8 && B >= 2) {
        typedef double hadi_d2s __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int q = 0; q < B / 2; q++) {
            hadi_d2s t;
            t.x = u[2 * q];
            t.y = u[2 * q + 1];
            __builtin_nontemporal_store(t, reinterpret_cast<hadi_d2s *>(row + q * 128 * G + 128 * half + 2 * lane));
        }
        return;
    }
#endif
    if constexpr (sizeof(T) == 4 && B >= 4) {
#pragma unroll
        for (int q = 0; q < B / 4; q++) {
            float4 t;
            t.x = (float)u[4 * q]; t.y = (float)u[4 * q + 1]; t.z = (float)u[4 * q + 2]; t.w = (float)u[4 * q + 3];
            *reinterpret_cast<float4 *>(row + q * 256 * G + 256 * half + 4 * lane) = t;
        }
    } else if constexpr (B == 1) {
        row[64 * half + lane] = (T)u[0];
    } else {
#pragma unroll
        for (int q = 0; q < B / 2; q++) {
            typename HadiPair<T>::type t;
            t.x = (T)u[2 * q];
            t.y = (T)u[2 * q + 1];
            *reinterpret_cast<typename HadiPair<T>::type *>(row + q * 128 * G + 128 * half + 2 * lane) = t;
        }
    }
}

// Vector stores hadi_put_block issues per lane for one row block: the counted vmcnt waits add this as the LOWER bound of
// the operations a row step puts behind a DMA batch (too high a count would let the wait pass with a DMA piece still in
// flight).  fp32 state at 4 or 8 nodes per lane stores QUADS (B/4 instructions), everything else pairs.
template <int B, class T>
HADI_DEV constexpr int hadi_put_block_stores() {
    return B == 1 ? 1 : (sizeof(T) == 4 && B >= 4) ? B / 4 : B / 2;
}

// Cache policy of the carry-over arrays R1 and C2: non-temporal, loads (a last use) and stores (read again two launches later,
// after 0.8 GB of other traffic) alike -- written or read with the default policy they take the place of the array the NEXT
// launch reads (Y, written by this pass) in the 256 MB memory-side cache.
#ifndef HADI_CS_POLICY
#define HADI_CS_POLICY " nt"
#endif
template <int B, int G>
HADI_DEV HADI_FORCEINLINE void hadi_put_block_nt(double *row, int half, int lane, const double (&u)[B]) {
#if defined(HADI_EMU)
    hadi_put_block<B, G>(row, half, lane, u);
#else
    typedef double hadi_d2 __attribute__((ext_vector_type(2)));
    if constexpr (B == 1) {
        __builtin_nontemporal_store(u[0], row + 64 * half + lane);
    } else {
#pragma unroll
        for (int q = 0; q < B / 2; q++) {
            hadi_d2 t;
            t.x = u[2 * q];
            t.y = u[2 * q + 1];
            __builtin_nontemporal_store(t, reinterpret_cast<hadi_d2 *>(row + q * 128 * G + 128 * half + 2 * lane));
        }
    }
#endif
}
template <int B, int G>
HADI_DEV HADI_FORCEINLINE void hadi_get_block_nt(const double *row, int half, int lane, double (&u)[B]) {
#if defined(HADI_EMU)
    hadi_get_block<B, G>(row, half, lane, u);
#else
    typedef double hadi_d2 __attribute__((ext_vector_type(2)));
    if constexpr (B == 1) {
        u[0] = __builtin_nontemporal_load(row + 64 * half + lane);
    } else {
#pragma unroll
        for (int q = 0; q < B / 2; q++) {
            const hadi_d2 t = __builtin_nontemporal_load(reinterpret_cast<const hadi_d2 *>(row + q * 128 * G + 128 * half + 2 * lane));
            u[2 * q] = t.x;
            u[2 * q + 1] = t.y;
        }
    }
#endif
}

// One v-row: explicit stage, Y0, A1 line solve, A2 right-hand side.  LAST = this is the v-row that
// carries b2 (hes_boundary_kernels.hpp:62-66); AMER adds lambda_bar (device_solver.hpp:325-331).
// `active` is wave-uniform; with G = 2 every wave of the block must call this (it contains a barrier).
// MODE 0: Douglas step.  MODE 1 / 2: predictor / corrector of Craig-Sneyd (solver.hpp:781-907).  With
//   Y1rhs = Y0 + theta dt (b1 e_n - (A1U + b1 e_{n-1})),  C2 = theta dt (b2 e_n - (A2U + b2 e_{n-1}))
// the corrector's A1 right-hand side is Y0~ + theta dt (...) = Y1rhs + dt/2 (A0 Y2 - A0 U): MODE 1 is a
// Douglas row step that also stores R1 = Y1rhs - dt/2 A0U and C2; MODE 2 takes its rows from Y2, forms
// R1 + dt/2 A0 Y2, runs the same A1 solve and adds C2 -- it never needs U, A1U or A2U again.
// AMER == 2: American in the P representation.  After the projection  U = max(P, U0),  lambda_bar = max(0, (U0 - P)/dt)
// with  P = U_bar - dt lambda_bar_old  (device_solver.hpp:358-372 rewritten), so ONE array -- P, stored where U is --
// carries both, provided the payoff U0 depends on s only (then it is a per-lane constant here).  The row pass rebuilds U on
// the five stencil rows and lambda_bar on row j from P; no lambda_bar array is read or written by the sweep.
template <int B, int G, int AMER, bool LAST, int MODE = 0, class T = double>
HADI_DEV HADI_FORCEINLINE void hadi_row_step(const HadiRowCtxT<T> &c, bool active, int j, const T *rm2, const T *rm1,
                                             const T *r0, const T *rp1, const T *rp2) {
    const int lane = c.lane, rowp = c.rowp, half = c.half;
    constexpr int c0slot = 64 * B * G;
    constexpr int NB = B - 1;
    const double dt = c.dt, thdt = c.thdt, qd = c.qd, half_rd = c.half_rd, e_nm1 = c.e_nm1, e_n = c.e_n;
    const bool first_half = (half == 0), last_half = (half == G - 1);
    // state that survives the exchange barrier
    double ys[B], ps[B], gs[B], A2U[B], b2v[B], r1v[B], c2v[B];
    double Ysol = 0.0, Ssol = 0.0, yout_c0 = 0.0;

    HADI_STAMP_DECL(c.stamp_acc_)
    if (active) {
        // (LDS, not global: an ordinary global load here would make hipcc drain the in-flight LDS-DMA
        // prefetch with vmcnt(0) at the start of every row)
        const double *rc = c.rowc + (size_t)(j - c.j0) * HADI_RCL;
        const double v = rc[RC_V];
        const double wm = rc[RC_WM], wz = rc[RC_WZ], wp = rc[RC_WP];
        const double a2l2 = rc[RC_L2], a2l1 = rc[RC_L1], a2m = rc[RC_M], a2u1 = rc[RC_U1], a2u2 = rc[RC_U2];
        const double b1val = rc[RC_B1VAL];
        const int b1raw = (int)rc[RC_B1COL];
        const bool b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;  // (two entries on one v-row: m2 > m1 only)
        const int b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
        // which (wave, lane, slot) holds the b1 node of this v-row
        const int b1e = b1col - 1;
        const int b1half = (b1col >= 1) ? b1e / (64 * B) : -1;
        const int b1el = b1e - b1half * 64 * B;
        const int b1lane = (b1col >= 1 && b1half == half) ? b1el / B : -1;
        const int b1r = b1el - (b1el / B) * B;

        // ---- column i = 0 (A0 and A1 rows are zero there; only A2 and the boundary act) ------------
        double c0 = (double)r0[c0slot];
        double c0m2 = (double)rm2[c0slot], c0m1 = (double)rm1[c0slot], c0p1 = (double)rp1[c0slot], c0p2 = (double)rp2[c0slot];
        double lamc0 = 0.0;
        if constexpr (AMER == 1) lamc0 = c.Li[(size_t)j * rowp + c0slot];
        if constexpr (AMER == 2) {
            const double pay0 = c.payrow[c0slot];
            lamc0 = fmax(0.0, (pay0 - c0) * c.inv_dt);
            c0 = fmax(c0, pay0); c0m2 = fmax(c0m2, pay0); c0m1 = fmax(c0m1, pay0); c0p1 = fmax(c0p1, pay0); c0p2 = fmax(c0p2, pay0);
        }
        const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c0 + a2u1 * c0p1 + a2u2 * c0p2;
        const double b1c0 = b1_at0 ? b1val : 0.0;
        const double b2c0 = LAST ? c.b2r[c0slot] : 0.0;
        const double a1c0 = -c.hr0 * c0;  // A1 row 0: empty for the call (hr0 = 0), the reaction term for the put
        double y0c0 = c0 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1 + lamc0);
        y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
        double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
        if constexpr (MODE == 1) {  // A0 is zero on i = 0: R1 = Y1rhs there
            if (lane == 0 && first_half) {
                c.R1i[(size_t)j * rowp + c0slot] = y0c0;
                c.C2i[(size_t)j * rowp + c0slot] = c2c0;
            }
        }
        if constexpr (MODE == 2) {
            y0c0 = c.R1i[(size_t)j * rowp + c0slot];
            c2c0 = c.C2i[(size_t)j * rowp + c0slot];
        }
        const double x0 = y0c0 * c.inv0;  // A1 row 0 is decoupled: the identity for the call (hes_a1_kernels.hpp:56-61)
        yout_c0 = x0 + c2c0;

        HADI_STAMP(0);  // row scalars + column 0
        // ---- explicit operators.  A0 = (s-derivative) o (v-derivative): first the v-combination
        // t = wm u(j-1) + wz u(j) + wp u(j+1) on the block and its two s-neighbours, then the B-weights.
        double u0[B], tt[B];
        double lam[B], pay[B];
        if constexpr (AMER == 2) hadi_get_block<B, G>(c.payrow, half, lane, pay);
        {
            double um[B], up[B], u2[B];
            hadi_get_block<B, G, T>(r0, half, lane, u0);
            hadi_get_block<B, G, T>(rm1, half, lane, um);
            hadi_get_block<B, G, T>(rp1, half, lane, up);
            if constexpr (AMER == 2) {
#pragma unroll
                for (int r = 0; r < B; r++) {
                    lam[r] = fmax(0.0, (pay[r] - u0[r]) * c.inv_dt);  // from the raw P of row j
                    if (lane == c.m1_lane && r == c.m1_r) lam[r] = 0.0;
                    u0[r] = fmax(u0[r], pay[r]);
                    um[r] = fmax(um[r], pay[r]);
                    up[r] = fmax(up[r], pay[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < B; r++) {
                tt[r] = wm * um[r] + wz * u0[r] + wp * up[r];
                A2U[r] = a2l1 * um[r] + a2m * u0[r] + a2u1 * up[r];
            }
            hadi_get_block<B, G, T>(rm2, half, lane, u2);
            if constexpr (AMER == 2) {
#pragma unroll
                for (int r = 0; r < B; r++) u2[r] = fmax(u2[r], pay[r]);
            }
#pragma unroll
            for (int r = 0; r < B; r++) A2U[r] = fma(a2l2, u2[r], A2U[r]);
            hadi_get_block<B, G, T>(rp2, half, lane, u2);
            if constexpr (AMER == 2) {
#pragma unroll
                for (int r = 0; r < B; r++) u2[r] = fmax(u2[r], pay[r]);
            }
#pragma unroll
            for (int r = 0; r < B; r++) A2U[r] = fma(a2u2, u2[r], A2U[r]);
        }
        const double cb1 = dt * e_nm1 + thdt * (e_n - e_nm1);
        const double b1l = (lane == b1lane) ? b1val * cb1 : 0.0;  // this row's b1 entry, in the lane that owns its node
        double u0L = (double)r0[c.posL], u0R = (double)r0[c.posR];
        double m1L = (double)rm1[c.posL], m1R = (double)rm1[c.posR], p1L = (double)rp1[c.posL], p1R = (double)rp1[c.posR];
        if constexpr (AMER == 2) {
            const double payL = c.payrow[c.posL], payR = c.payrow[c.posR];
            u0L = fmax(u0L, payL); m1L = fmax(m1L, payL); p1L = fmax(p1L, payL);
            u0R = fmax(u0R, payR); m1R = fmax(m1R, payR); p1R = fmax(p1R, payR);
        }
        const double tL = wm * m1L + wz * u0L + wp * p1L;
        const double tR = wm * m1R + wz * u0R + wp * p1R;

        HADI_STAMP(1);  // LDS rows -> tt, A2U
        if constexpr (AMER == 1) hadi_get_block<B, G>(c.Li + (size_t)j * rowp, half, lane, lam);
        if constexpr (LAST) hadi_get_block<B, G>(c.b2r, half, lane, b2v);
        if constexpr (MODE == 2) {
            hadi_get_block_nt<B, G>(c.R1i + (size_t)j * rowp, half, lane, r1v);
            hadi_get_block_nt<B, G>(c.C2i + (size_t)j * rowp, half, lane, c2v);
        }

        // ---- Y0 (device_solver.hpp:236-250) fused with the forward sweep of the in-lane Thomas ----------
        //   x[r] = ys[r] - XL*ps[r] - X*gs[r],  XL = interface unknown of lane-1, X = own x[B-1]
        // The central FD weights follow from sum_k beta_s = sum_k delta_s = 0: B0 = -(Bm+Bp), D0 = -(Dm+Dp).
        hadi_set_prio(1);  // see hadi_set_prio
        double Bm[B], Bp[B], Dm[B], Dp[B];
        hadi_get_block<B, G>(c.coef + 0 * 64 * B * G, half, lane, Bm);
        hadi_get_block<B, G>(c.coef + 1 * 64 * B * G, half, lane, Bp);
        hadi_get_block<B, G>(c.coef + 2 * 64 * B * G, half, lane, Dm);
        hadi_get_block<B, G>(c.coef + 3 * 64 * B * G, half, lane, Dp);
        double iu[B], cp[B];
        double il_last = 0.0, im_last = 1.0, d_last = 0.0;
        double il_moved = 0.0;  // the i = 1 row's coupling to x_0 once it has been moved to the right-hand side
#pragma unroll
        for (int r = 0; r < B; r++) {
            const double uL = (r == 0) ? u0L : u0[r == 0 ? 0 : r - 1];
            const double uR = (r == B - 1) ? u0R : u0[r == B - 1 ? r : r + 1];
            const double tl = (r == 0) ? tL : tt[r == 0 ? 0 : r - 1];
            const double tr = (r == B - 1) ? tR : tt[r == B - 1 ? r : r + 1];
            const double lo = fma(v, Dm[r], qd * Bm[r]);
            const double up = fma(v, Dp[r], qd * Bp[r]);
            const double mn = -((lo + up) + half_rd);  // = -(v (Dm + Dp) + q (Bm + Bp) + r_d / 2)
            const double A1U = lo * uL + mn * u0[r] + up * uR;
            const double A0U = Bm[r] * tl - (Bm[r] + Bp[r]) * tt[r] + Bp[r] * tr;
            // Y0 = U + dt (A0U + A1U + A2U + b e_{n-1} [+ lambda]) + theta dt (b1 e_n - (A1U + b1 e_{n-1})); the b1
            // entry of this v-row (a single node) contributes b1 * cb1, cb1 = dt e_{n-1} + theta dt (e_n - e_{n-1})
            double S = A0U + A1U + A2U[r];
            if constexpr (LAST) S += b2v[r] * e_nm1;
            if constexpr (AMER != 0) S += lam[r];
            double y = fma(dt, S, u0[r]);
            y = fma(-thdt, A1U, y);
            y = fma(b1l, (r == b1r) ? 1.0 : 0.0, y);  // wave-uniform selector: one FMA with a scalar operand
            if constexpr (MODE == 1) r1v[r] = fma(-0.5 * dt, A0U, y);
            if constexpr (MODE == 2) y = fma(0.5 * dt, A0U, r1v[r]);  // A0U is A0 applied to Y2 here
            double il = -thdt * lo;
            const double im = 1.0 - thdt * mn;
            iu[r] = -thdt * up;
            if (r == 0 && lane == 0 && first_half) {  // x_0 is known: move it to the right-hand side
                y -= il * x0;
                il_moved = il;
                il = 0.0;
            }
            if (r < NB) {
                // normalised rows (x[r] + cp[r] x[r+1] = ys[r] - ps[r] XL): the back substitution is then one FMA per vector
                if (r == 0) {
                    const double inv = hadi_rcp(im);
                    cp[0] = iu[0] * inv;
                    ys[0] = y * inv;
                    ps[0] = il * inv;
                } else {
                    const double inv = hadi_rcp(fma(-il, cp[r - 1], im));
                    cp[r] = iu[r] * inv;
                    ys[r] = fma(-il, ys[r - 1], y) * inv;
                    ps[r] = -(il * ps[r - 1]) * inv;
                }
            } else {
                il_last = il;
                im_last = im;
                d_last = y;
            }
        }
        (void)il_moved;
        HADI_STAMP(2);  // coefficients + Y0 + forward Thomas
        // reduced (interface) row of this lane:  ra*X(l-1) + rb*X(l) + rcc*X(l+1) = rf [- rs * boundary value]
        double ra, rb, rcc, rf, rs = 0.0;
        const bool edge_hi = (G > 1) && !last_half && lane == 63;  // next node belongs to the other wave
        const bool edge_lo = (G > 1) && !first_half && lane == 0;  // previous node belongs to the other wave
        if constexpr (NB > 0) {
            gs[NB - 1] = cp[NB - 1];
#pragma unroll
            for (int r = NB - 2; r >= 0; r--) {
                ys[r] = fma(-cp[r], ys[r + 1], ys[r]);
                ps[r] = fma(-cp[r], ps[r + 1], ps[r]);
                gs[r] = -cp[r] * gs[r + 1];
            }
            double p0n = hadi_lane_next(ps[0]), g0n = hadi_lane_next(gs[0]), y0n = hadi_lane_next(ys[0]);
            if (edge_hi) { p0n = 0.0; g0n = 0.0; y0n = 0.0; }
            ra = -il_last * ps[NB - 1];
            rb = im_last - il_last * gs[NB - 1] - iu[B - 1] * p0n;
            rcc = -iu[B - 1] * g0n;
            rf = d_last - il_last * ys[NB - 1] - iu[B - 1] * y0n;
        } else {
            ra = il_last;
            rb = im_last;
            rcc = iu[0];
            rf = d_last;
        }
        if constexpr (G > 1) {
            if (edge_hi) { rs = iu[B - 1]; rcc = 0.0; }  // couples to t = first node of the other half
            if (edge_lo) { rs = ra; ra = 0.0; }          // couples to the last node of the other half
        }
        HADI_STAMP(3);  // backward Thomas + reduced row
        // ---- parallel cyclic reduction over the 64 interface unknowns (normalised rows) -------------
        // Lanes without a partner at distance s have ra == 0 (left) / rcc == 0 (right) by induction, so the
        // (wrapped) values they fetch are multiplied by zero: no lane masks are needed.
        if constexpr (NB == 0) {
            // One node per lane (m1 <= 64: the reference's calibration grids): the cyclic reduction IS the whole line solve,
            // so two nodes a tiny interval apart (S_0 inserted 7e-6 beside a node: off-diagonals of 1e7 against a row sum of
            // ~1) are two of its unknowns.  In the plain update the new diagonal 1 - a cL - c aR is a difference of numbers
            // that agree to 7 digits; the two nodes come out with independent errors of cond * eps each, and the NEXT
            // step multiplies their difference by the 1e7 coupling again (found by the extended-precision adjudicator,
            // oracle/heston_oracle_xp.c: fuzz seed 5 case 279, field error 1.7e-7 against 4.5e-10 for the Thomas sweep
            // of the reference).  Carrying every row's EXCESS d = 1 + a + c (diagonal dominance; known analytically,
            // 1 + theta dt r_d / 2 before normalisation) removes the cancellation: with cL = dL - 1 - aL, aR = dR - 1 - cR
            //   new excess   e  = d - a dL - c dR          (for an M-matrix row: a sum of non-negative terms)
            //   new diagonal bn = e + a aL + c cR           (likewise)
            // Same number of cross-lane fetches as the plain update (dL, dR replace cL, aR), three more VALU operations
            // per level.  The algebra holds for any signs; only the no-cancellation property needs a, c <= 0.
            hadi_set_prio(3);
            double rd = (1.0 + thdt * half_rd) - il_moved;  // row sum il + im + iu of I - theta dt A1 (il of the first node moved out)
            rb = rd - ra - rcc;                              // diagonal from the off-diagonals and the excess
            const double rinv0 = hadi_rcp(rb);
            ra *= rinv0;
            rcc *= rinv0;
            rf *= rinv0;
            rd *= rinv0;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const int up_lane = (lane - s) & 63, dn_lane = (lane + s) & 63;
                double aL, dL, fL, cR, dR, fR;
                if (s == 1) {
                    aL = hadi_lane_prev(ra); dL = hadi_lane_prev(rd); fL = hadi_lane_prev(rf);
                    cR = hadi_lane_next(rcc); dR = hadi_lane_next(rd); fR = hadi_lane_next(rf);
                } else if (s == 32) {  // lane - 32 and lane + 32 are the same lane (mod 64)
                    aL = hadi_lane_get(ra, up_lane); cR = hadi_lane_get(rcc, up_lane);
                    dL = dR = hadi_lane_get(rd, up_lane); fL = fR = hadi_lane_get(rf, up_lane);
                } else {
                    aL = hadi_lane_get(ra, up_lane); dL = hadi_lane_get(rd, up_lane); fL = hadi_lane_get(rf, up_lane);
                    cR = hadi_lane_get(rcc, dn_lane); dR = hadi_lane_get(rd, dn_lane); fR = hadi_lane_get(rf, dn_lane);
                }
                const double e = fma(-rcc, dR, fma(-ra, dL, rd));
                const double bn = fma(rcc, cR, fma(ra, aL, e));
                const double rn = hadi_rcp(bn);
                rf = fma(-rcc, fR, fma(-ra, fL, rf)) * rn;
                if (s < 32) {  // the last level only needs the right-hand side
                    const double an = -(ra * aL) * rn;
                    const double cn = -(rcc * cR) * rn;
                    ra = an;
                    rcc = cn;
                    rd = e * rn;
                }
            }
        } else {
            hadi_set_prio(3);
            const double rinv0 = hadi_rcp(rb);
            ra *= rinv0;
            rcc *= rinv0;
            rf *= rinv0;
            if constexpr (G > 1) rs *= rinv0;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const int up_lane = (lane - s) & 63, dn_lane = (lane + s) & 63;
                double aL, cL, fL, aR, cR, fR;
                if (s == 1) {  // (constant after unrolling) the first level's neighbours are one lane away
                    aL = hadi_lane_prev(ra); cL = hadi_lane_prev(rcc); fL = hadi_lane_prev(rf);
                    aR = hadi_lane_next(ra); cR = hadi_lane_next(rcc); fR = hadi_lane_next(rf);
                } else if (s == 32) {  // lane - 32 and lane + 32 are the same lane (mod 64): one fetch serves both sides
                    aL = aR = hadi_lane_get(ra, up_lane); cL = cR = hadi_lane_get(rcc, up_lane); fL = fR = hadi_lane_get(rf, up_lane);
                } else {
                    aL = hadi_lane_get(ra, up_lane); cL = hadi_lane_get(rcc, up_lane); fL = hadi_lane_get(rf, up_lane);
                    aR = hadi_lane_get(ra, dn_lane); cR = hadi_lane_get(rcc, dn_lane); fR = hadi_lane_get(rf, dn_lane);
                }
                const double bn = fma(-rcc, aR, fma(-ra, cL, 1.0));
                const double rn = hadi_rcp(bn);
                rf = fma(-rcc, fR, fma(-ra, fL, rf)) * rn;
                if constexpr (G > 1) {
                    const double sL = (s == 1) ? hadi_lane_prev(rs) : hadi_lane_get(rs, up_lane);
                    const double sR = (s == 1) ? hadi_lane_next(rs) : (s == 32) ? sL : hadi_lane_get(rs, dn_lane);
                    rs = fma(-rcc, sR, fma(-ra, sL, rs)) * rn;
                }
                if (s < 32) {  // the last level only needs the right-hand sides
                    const double an = -(ra * aL) * rn;
                    const double cn = -(rcc * cR) * rn;
                    ra = an;
                    rcc = cn;
                }
            }
        }
        hadi_set_prio(0);
        HADI_STAMP(4);  // PCR
        Ysol = rf;
        Ssol = rs;
        if constexpr (G > 1) {
            // X(l) = Ysol - bv * Ssol with bv the other half's adjacent node.  Publish what the 2x2 needs:
            //   low half, lane 63:  x_hi = A - t*Bc              (A = Ysol, Bc = Ssol; x_hi = its own X)
            //   high half, lane 0:  t = C - x_hi*D   (t = its first node = ys0 - XL ps0 - X gs0, XL = x_hi)
            if (edge_hi) {
                c.xch[8 * c.wrow + 0] = Ysol;
                c.xch[8 * c.wrow + 1] = Ssol;
            }
            if (edge_lo) {
                if constexpr (NB > 0) {
                    c.xch[8 * c.wrow + 2] = ys[0] - Ysol * gs[0];
                    c.xch[8 * c.wrow + 3] = ps[0] - Ssol * gs[0];
                } else {
                    c.xch[8 * c.wrow + 2] = Ysol;
                    c.xch[8 * c.wrow + 3] = Ssol;
                }
            }
        }
    }
    if constexpr (G > 1) {
#if defined(HADI_EMU) || defined(HADI_BLOCK_EXCHANGE)
        __syncthreads();
#else
        // Rendezvous of the TWO wavefronts of this v-row only (the other rows of the block run on): each publishes a token
        // behind its two values (same lane, so the LDS unit sees data before flag) and polls the partner's.  Both are
        // resident wavefronts of one block and `active` is the same for both, so the partner always arrives; the loop-top
        // barrier of the next iteration separates this exchange from the next use of the slots.  The poll is bounded so
        // that a logic error can never hang the GPU; running out of polls is reported through the handle's error word
        // (hadi_report) and fails the call.
        if (active) {
            int *flags = reinterpret_cast<int *>(c.xch + 8 * c.wrow + 4);
            const int token = j + 1;
            const bool publisher = (!last_half && lane == 63) || (!first_half && lane == 0);  // the lanes that wrote the values
            const bool withhold = (c.debug & HADI_DEBUG_WITHHOLD_TOKEN) && half == 1 && j == 1;  // (test hook)
            if (publisher && !withhold) __hip_atomic_store(flags + half, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            int guard = 0;
            const int polls = HADI_RENDEZVOUS_POLLS(c.debug);
            while (__hip_atomic_load(flags + (1 - half), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != token && ++guard < polls)
                __builtin_amdgcn_s_sleep(1);
            if (guard >= polls && lane == 0) hadi_report(c.err, HADI_DEVERR_RENDEZVOUS);
        }
#endif
    }
    if (active) {
        double X = Ysol, XL;
        if constexpr (G > 1) {
            const double A = c.xch[8 * c.wrow + 0], Bc = c.xch[8 * c.wrow + 1];
            const double Cc = c.xch[8 * c.wrow + 2], Dd = c.xch[8 * c.wrow + 3];
            const double xhi = (A - Bc * Cc) * hadi_rcp(1.0 - Bc * Dd);  // last node of the low half (reciprocal + Newton step as in the line solves: the IEEE division sequence is 12 dependent instructions on the pair's critical path)
            const double tlo = Cc - Dd * xhi;                    // first node of the high half
            X = Ysol - (first_half ? tlo : xhi) * Ssol;
            XL = hadi_lane_prev(X);
            if (lane == 0) XL = first_half ? 0.0 : xhi;
        } else {
            XL = hadi_lane_prev(X);
        }
        // ---- Y1 -> right-hand side of the A2 solve (device_solver.hpp:254-260) and store ----------
        double yo[B];
#pragma unroll
        for (int r = 0; r < B; r++) {
            double x;
            if (r < NB) x = ys[r] - XL * ps[r] - X * gs[r];
            else x = X;
            double corr;
            if constexpr (MODE == 2) corr = c2v[r];
            else if constexpr (LAST) corr = thdt * (b2v[r] * e_n - (A2U[r] + b2v[r] * e_nm1));
            else corr = -thdt * A2U[r];
            yo[r] = x + corr;
            if constexpr (MODE == 1) c2v[r] = corr;
        }
        if constexpr (MODE == 1) {
            hadi_put_block_nt<B, G>(c.R1i + (size_t)j * rowp, half, lane, r1v);
            hadi_put_block_nt<B, G>(c.C2i + (size_t)j * rowp, half, lane, c2v);
        }
        hadi_put_block<B, G, T>(c.Yi + (size_t)j * rowp, half, lane, yo);
        if (lane == 0 && first_half) c.Yi[(size_t)j * rowp + c0slot] = (T)yout_c0;
        HADI_STAMP(5);  // final correction + store
    }
}

// Counted wait: at most `n` of this wavefront's youngest vector-memory operations may still be in flight.
HADI_DEV HADI_FORCEINLINE void hadi_wait_vmcnt(int n) {
#if defined(HADI_STRICT_VMCNT) && !defined(HADI_EMU)
    // Checking build (libhadi_strict.so, tests only): every counted wait becomes a full drain.  The counted waits rest on
    // hand-kept instruction counts (DMA pieces per row, stores per row); if a compiler change ever broke that bookkeeping
    // the product build would read stale ring rows while this build stays right -- tests/test_gpu_parity.py compares the
    // two bit for bit on every strip / ring shape.
    (void)n;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif !defined(HADI_EMU)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;  // n >= 25: stricter than asked, still safe
    }
#else
    (void)n;
#endif
}

// NG = row tiles handled by one block (each by its own group of W*G wavefronts with its own LDS ring; the
// s-coefficient arrays are shared), PD = prefetch depth in iterations: the ring holds (PD+1)*W + 4 rows.
// Every shape runs NG = 1 (hadi_plan.h: two groups behind one barrier measured slower); PD = 1 at 8 nodes per lane
// (two 4-wave blocks per CU), 2 below.  Large batches at 8 nodes per lane use hadi_pass_a_strip instead.
// T = float: fp32-state sweep (a.U / a.Y then point to float arrays of the same element layout; European Douglas only).
template <int B, int G, int W, int NG, int PD, int AMER, int MODE = 0, class T = double>
// Minimum blocks per CU of the shared-ring kernel at 4 nodes per lane: 3 (170 VGPRs) -- at 4 (128 VGPRs) the American
// variants spill into scratch inside the row loop (measured, 256x128 x512 American puts: 0.0966 -> 0.0942 ms per launch)
#ifndef HADI_RING_OCC_B4
#define HADI_RING_OCC_B4 3
#endif
__global__ void __launch_bounds__(64 * W * G * NG, (B >= 8 ? 2 : B == 4 ? HADI_RING_OCC_B4 : 4)) hadi_pass_a(HadiSweepArgs a, int n) {
    static_assert(sizeof(T) == 8 || (!AMER && MODE == 0), "the fp32-state sweep covers the European Douglas step only");
    HADI_DYN_SMEM(double, smem);
    constexpr int RING = (PD + 1) * W + 4;
    constexpr int NT = 64 * W * G * NG;
    const int lane = threadIdx.x & 63;
    const int wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    const int grp = wave / (W * G), wv = wave - grp * (W * G);
    const int wrow = wv / G, half = wv - wrow * G;
    const int tblocks = (a.ntiles + NG - 1) / NG;  // blocks per instance
    const int total = a.n_inst * tblocks;
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);
    if (logical >= total) return;
    const int inst = logical / tblocks, tb = logical - inst * tblocks;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;
    const int nrows = a.L.nrows, npad = a.L.nrows_pad, rowp = a.L.rowp;
    const int tile = tb * NG + grp;
    const int j0 = tile * a.R;  // may be >= nrows for the last block's spare group: that group only joins barriers
    const int j1 = (j0 + a.R < nrows) ? j0 + a.R : nrows;

    HadiRowCtxT<T> c;
    c.lane = lane;
    c.half = half;
    c.wrow = wrow;
    c.rowp = rowp;
    c.dt = ip.dt; c.thdt = ip.thdt; c.qd = ip.q; c.half_rd = ip.half_rd;
    c.hr0 = ip.hr0; c.inv0 = 1.0 / (1.0 + ip.thdt * ip.hr0);
    c.e_nm1 = exp(ip.bc_rate * ip.dt * (n - 1));  // device_solver.hpp:238
    c.e_n = exp(ip.bc_rate * ip.dt * n);          // device_solver.hpp:246
    const T *__restrict__ Ub = reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride;
    c.Yi = reinterpret_cast<T *>(a.Y) + (size_t)inst * a.L.inst_stride;
    c.Li = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride : nullptr;
    c.b2r = a.b2row + (size_t)inst * rowp;
    c.R1i = MODE ? a.R1 + (size_t)inst * a.L.inst_stride : nullptr;
    c.C2i = MODE ? a.C2 + (size_t)inst * a.L.inst_stride : nullptr;
    c.j0 = j0;
    c.err = a.err; c.debug = a.debug;
    constexpr int c0slot = 64 * B * G;
    // storage positions of the s-neighbours of this lane's block (node before its first, node after its
    // last).  Before i = 1 comes the i = 0 slot; after the row's last node comes a pad slot (always 0).
    {
        const int ifirst = 1 + 64 * B * half + B * lane;
        if constexpr (sizeof(T) == 4) {
            c.posL = hadi_pos_f32(B, G, ifirst - 1);
            c.posR = (ifirst + B <= 64 * B * G) ? hadi_pos_f32(B, G, ifirst + B) : c0slot + 1;
        } else {
            c.posL = hadi_pos(B, G, ifirst - 1);
            c.posR = (ifirst + B <= 64 * B * G) ? hadi_pos(B, G, ifirst + B) : c0slot + 1;
        }
    }

    // LDS: [NG rings of RING rows of T] [4 coefficient arrays of 64*B*G] [NG*W*8 exchange] [NG compact row tables]
    T *ring = reinterpret_cast<T *>(smem) + (size_t)grp * RING * rowp;
    double *coef = reinterpret_cast<double *>(reinterpret_cast<T *>(smem) + (size_t)NG * RING * rowp);
    {
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 64 * B * G;
        for (int e = threadIdx.x; e < 4 * 64 * B * G; e += NT) coef[e] = sc[e];
    }
    c.coef = coef;
    c.xch = coef + 4 * 64 * B * G + grp * 8 * W;  // per v-row: 4 exchange values + the two rendezvous tokens
    if (threadIdx.x < 8 * W * NG) coef[4 * 64 * B * G + threadIdx.x] = 0.0;  // (tokens start at 0; the first loop barrier publishes this)
    {
        double *rtab = coef + 4 * 64 * B * G + NG * 8 * W + (size_t)grp * a.R * HADI_RCL;
        const double *__restrict__ rg = a.rowc + ((size_t)inst * nrows + j0) * HADI_RC;
        const int tl = threadIdx.x - grp * 64 * W * G;
        for (int e = tl; e < (j1 - j0) * HADI_RCL; e += 64 * W * G) rtab[e] = rg[(e / HADI_RCL) * HADI_RC + e % HADI_RCL];
        c.rowc = rtab;
        c.payrow = nullptr; c.inv_dt = 0.0; c.m1_lane = -1; c.m1_r = -1;
        if constexpr (AMER == 2) {  // payoff row (v-row 0 of the packed payoff; it depends on s only) after the tables
            double *prow = coef + 4 * 64 * B * G + NG * 8 * W + (size_t)NG * a.R * HADI_RCL;
            const double *__restrict__ pg = a.U0 + (size_t)inst * a.L.inst_stride;
            for (int e = threadIdx.x; e < rowp; e += NT) prow[e] = pg[e];
            c.payrow = prow;
            c.inv_dt = 1.0 / ip.dt;
            const int e1 = a.L.m1 - 1;  // node i = m1 is element m1-1 of the row's 64*B*G interior nodes
            if (e1 / (64 * B) == half) {
                c.m1_lane = (e1 - half * 64 * B) / B;
                c.m1_r = (e1 - half * 64 * B) % B;
            }
        }
    }

    const int iters = (j1 > j0) ? (j1 - j0 + W - 1) / W : 0;  // this group's iterations
    const int iters_all = (a.R + W - 1) / W;                   // every group of the block runs this many barriers
    auto slot = [&](int jj) { return ring + (size_t)((jj + 4 * RING) % RING) * rowp; };
    // fetch returns the number of vector-memory instructions it issued
    auto fetch = [&](int jj) -> int {
        const bool exists = jj >= 0 && jj < npad;
        if constexpr ((64 * B * G + HADI_ROW_PAD(B, (int)sizeof(T))) % (16 / (int)sizeof(T)) == 0)
            hadi_row_to_lds_fixed<B, T, G>(Ub + (ptrdiff_t)jj * rowp, slot(jj), lane, exists);
        else
            hadi_row_to_lds(Ub + (size_t)jj * rowp, slot(jj), rowp, lane, exists);
        return exists ? hadi_row_dma_count<T>(rowp) : 0;
    };
    // prologue: rows of iterations 0 .. PD-1
    if (iters > 0)
        for (int rr = wv; rr < PD * W + 4; rr += W * G) fetch(j0 - 2 + rr);

    // Vector-memory operations retire in issue order.  ya[k] = (lower bound of the) number of operations this
    // wavefront issued after the DMA batch that iteration it+k needs, so hadi_wait_vmcnt(ya[0]) retires that
    // batch and leaves younger batches and result stores in flight.
    int ya[PD];
#pragma unroll
    for (int k = 0; k < PD; k++) ya[k] = 0;
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c.stamp_acc_ = stamp_store_;
#endif
    HADI_STAMP_DECL(stamp_store_)
    HADI_STAMP(8);  // prologue
    for (int it = 0; it < iters_all; it++) {
        const int J = j0 + it * W;
        hadi_wait_vmcnt(ya[0]);
        __syncthreads();  // this iteration's rows have landed; everyone is done with the rows replaced below
        HADI_STAMP(9);  // barrier wait (incl. DMA drain)
        int z = 0;
        if (it + PD < iters && wv < W) z = fetch(J + PD * W + 2 + wv);
#pragma unroll
        for (int k = 0; k + 1 < PD; k++) ya[k] = ya[k + 1] + z;
        ya[PD - 1] = 0;
        if constexpr (PD == 1) ya[0] = 0;
        const int j = J + wrow;
        const bool active = it < iters && j < j1;
        if constexpr (G == 1) {
            if (!active) continue;
        }
        if (j == nrows - 1)
            hadi_row_step<B, G, AMER, true, MODE, T>(c, active, j, slot(j - 2), slot(j - 1), slot(j), slot(j + 1), slot(j + 2));
        else
            hadi_row_step<B, G, AMER, false, MODE, T>(c, active, j, slot(j - 2), slot(j - 1), slot(j), slot(j + 1), slot(j + 2));
        if (active) {  // B/2 (one for B = 1) vector stores of the block; the i = 0 store is not counted (lower bound)
#pragma unroll
            for (int k = 0; k < PD; k++) ya[k] += hadi_put_block_stores<B, T>();
        }
        HADI_STAMP(10);  // whole row step (+ fetch issue)
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (lane == 0)
        for (int k = 0; k < 12; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
}

