// hadi_k_seq.h -- sequential passes for grids beyond the streaming kernels (hadi_pass_a_seq: m1 > 1024; hadi_pass_b_seq: m2 > 527).
// Part of libhadi's device code: include through hadi_kernels.h (which fixes the order).
#pragma once

// ------------------------------------------------------------------------------------------------
// Grids beyond the streaming kernels' shapes -- the reference bounds a grid by its total size only
// (src/perfomance_test.cpp:62).  Two sequential passes in the reference's own mapping (hes_a1_kernels.hpp:139-161: one
// thread per v-row; hes_a2_shuffled_kernels.hpp:243-299: one thread per s-column), correct for ANY shape, far from the
// roofline:
//   hadi_pass_a_seq  (m1 > 1024)   lane <-> v-row walks i = 1 .. m1 exactly as hadi_small_seq_kernel does in LDS, here on the
//                                  global arrays: explicit operators from a sliding window of three columns, Y0, forward
//                                  Thomas with the pivot recomputed on the fly; g_i goes to Y, the multiplier c'_i to a
//                                  scratch array (the handle's Craig-Sneyd buffer R1), the way back runs on Y in place.
//                                  Rows are kept in natural order (hadi_pick_shape: B = 1, slot of node i >= 1 is i - 1).
//   hadi_pass_b_seq  (m2 > 527)    lane <-> storage column (coalesced): forward sweep with the UNCHUNKED factorisation
//                                  (layout: one chunk of all rows, hadi_setup_instance) written over Y, backward sweep into U
//                                  with the Ikonen-Toivanen projection for American sweeps (explicit (U, lambda_bar) pair).
// European, dividend and American sweeps, call and put, fp64 state; the P representation, Craig-Sneyd and the fp32 state stay
// with the streaming kernels (the host keeps them off these shapes).
template <int AMER>
__global__ void __launch_bounds__(64) hadi_pass_a_seq(HadiSweepArgs a, int n) {
    typedef double T;
    static_assert(AMER == 0 || AMER == 1, "explicit (U, lambda_bar) pair only");
    const int lane = threadIdx.x;
    const int nrows = a.L.nrows, rowp = a.L.rowp, m1 = a.L.m1, nslot = 64 * a.L.B * a.L.G;
    const int jblocks = (nrows + 63) / 64;
    if ((int)blockIdx.x >= a.n_inst * jblocks) return;
    const int inst = blockIdx.x / jblocks, jb = blockIdx.x - inst * jblocks;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;
    const int j = jb * 64 + lane;
    const bool act = j < nrows, last = (j == nrows - 1);
    const int jr = act ? j : 0;  // (idle lanes walk row 0 and store nothing)
    const T *__restrict__ Ub = reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride;
    T *__restrict__ Yr = reinterpret_cast<T *>(a.Y) + (size_t)inst * a.L.inst_stride + (size_t)jr * rowp;
    double *__restrict__ Wr = a.R1 + (size_t)inst * a.L.inst_stride + (size_t)jr * rowp;  // c'_i at slot i - 1
    const double *__restrict__ Lr = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride + (size_t)jr * rowp : nullptr;
    const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * nslot;  // Bm, Bp, Dm, Dp of node i at [k * nslot + i - 1]
    const double *__restrict__ b2g = a.b2row + (size_t)inst * rowp;
    double v, wm, wz, wp, a2l2, a2l1, a2m, a2u1, a2u2, b1val;
    int b1col;
    bool b1_at0;
    {
        const double *__restrict__ rc = a.rowc + ((size_t)inst * nrows + jr) * HADI_RC;
        v = rc[RC_V]; wm = rc[RC_WM]; wz = rc[RC_WZ]; wp = rc[RC_WP];
        a2l2 = rc[RC_L2]; a2l1 = rc[RC_L1]; a2m = rc[RC_M]; a2u1 = rc[RC_U1]; a2u2 = rc[RC_U2];
        b1val = rc[RC_B1VAL];
        const int b1raw = (int)rc[RC_B1COL];
        b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;
        b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
    }
    const double dt = ip.dt, thdt = ip.thdt, qd = ip.q, half_rd = ip.half_rd;
    const double inv0 = 1.0 / (1.0 + ip.thdt * ip.hr0);
    // v-neighbours clamped to the grid: a clamped row only ever meets a zero weight (hadi_small_seq_kernel)
    const T *pm2 = Ub + (size_t)(jr >= 2 ? jr - 2 : 0) * rowp, *pm1 = Ub + (size_t)(jr >= 1 ? jr - 1 : 0) * rowp;
    const T *pr0 = Ub + (size_t)jr * rowp;
    const T *pp1 = Ub + (size_t)(jr + 1 < nrows ? jr + 1 : nrows - 1) * rowp, *pp2 = Ub + (size_t)(jr + 2 < nrows ? jr + 2 : nrows - 1) * rowp;
    // node i of a row: slot nslot for i = 0, slot i - 1 for 1 <= i <= m1, zero beyond (the s-neighbour of the last node)
    auto at = [&](const T *row, int i) -> double { return i == 0 ? (double)row[nslot] : (i <= m1 ? (double)row[i - 1] : 0.0); };
    auto b2at = [&](int i) -> double { return (last && i <= m1) ? (i == 0 ? b2g[nslot] : b2g[i - 1]) : 0.0; };
    const double e_nm1 = exp(ip.bc_rate * ip.dt * (n - 1));  // device_solver.hpp:238
    const double e_n = exp(ip.bc_rate * ip.dt * n);          // device_solver.hpp:246
    const double b1l = b1val * (dt * e_nm1 + thdt * (e_n - e_nm1));
    // column i = 0 (A0 and A1 rows are zero there; only A2 and the boundary act)
    const double c0m2 = at(pm2, 0), c0m1 = at(pm1, 0), c00 = at(pr0, 0), c0p1 = at(pp1, 0), c0p2 = at(pp2, 0);
    double r_m2 = at(pm2, 1), r_m1 = at(pm1, 1), r_0 = at(pr0, 1), r_p1 = at(pp1, 1), r_p2 = at(pp2, 1);
    double yout_c0, x0;
    {
        const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c00 + a2u1 * c0p1 + a2u2 * c0p2;
        const double b1c0 = b1_at0 ? b1val : 0.0;
        const double b2c0 = b2at(0);
        const double lamc0 = (AMER == 1) ? Lr[nslot] : 0.0;
        const double a1c0 = -ip.hr0 * c00;
        double y0c0 = c00 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1 + lamc0);
        y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
        const double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
        x0 = y0c0 * inv0;
        yout_c0 = x0 + c2c0;
    }
    double u_prev = c00, u_cur = r_0;
    double t_prev = wm * c0m1 + wz * c00 + wp * c0p1;
    double t_cur = wm * r_m1 + wz * r_0 + wp * r_p1;
    double a2u_cur = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
    double b2c = b2at(1);
    double corr_cur = thdt * (b2c * e_n - (a2u_cur + b2c * e_nm1));
    r_m2 = at(pm2, 2); r_m1 = at(pm1, 2); r_0 = at(pr0, 2); r_p1 = at(pp1, 2); r_p2 = at(pp2, 2);
    double cp_prev = 0.0, ys_prev = x0;  // x_0 is known: with ys_0 = x_0 and c'_0 = 0 the general step moves it to the right-hand side
    for (int i = 1; i <= m1; i++) {
        const double u_next = r_0;
        const double t_next = wm * r_m1 + wz * r_0 + wp * r_p1;
        const double a2u_next = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
        const int inx = i + 2;
        r_m2 = at(pm2, inx); r_m1 = at(pm1, inx); r_0 = at(pr0, inx); r_p1 = at(pp1, inx); r_p2 = at(pp2, inx);
        const double Bm = sc[0 * nslot + i - 1], Bp = sc[1 * nslot + i - 1], Dm = sc[2 * nslot + i - 1], Dp = sc[3 * nslot + i - 1];
        const double lo = fma(v, Dm, qd * Bm);
        const double up = fma(v, Dp, qd * Bp);
        const double mn = -((lo + up) + half_rd);
        const double A1U = lo * u_prev + mn * u_cur + up * u_next;
        const double A0U = Bm * t_prev - (Bm + Bp) * t_cur + Bp * t_next;
        double S = A0U + A1U + a2u_cur;  // Y0 = U + dt (A0U + A1U + A2U + b e_{n-1} [+ lambda_bar]) + ..., device_solver.hpp:236-250
        S += b2c * e_nm1;
        if constexpr (AMER == 1) S += Lr[i - 1];
        double y = fma(dt, S, u_cur);
        y = fma(-thdt, A1U, y);
        y += (i == b1col) ? b1l : 0.0;
        const double il = -thdt * lo;
        const double im = 1.0 - thdt * mn;
        const double iu = -thdt * up;
        const double inv = hadi_rcp(fma(-il, cp_prev, im));
        const double cp = iu * inv;
        const double ys = fma(-il, ys_prev, y) * inv;
        const double b2n = b2at(i + 1);
        const double corr_next = thdt * (b2n * e_n - (a2u_next + b2n * e_nm1));
        if (act) {
            Yr[i - 1] = ys + corr_cur + cp * corr_next;  // g_i (c'_{m1} = 0: the row ends there)
            Wr[i - 1] = cp;
        }
        u_prev = u_cur; u_cur = u_next;
        t_prev = t_cur; t_cur = t_next;
        a2u_cur = a2u_next; corr_cur = corr_next; b2c = b2n;
        cp_prev = cp; ys_prev = ys;
    }
    if (act) {  // back substitution on the output itself: Y_i = g_i - c'_i Y_{i+1}
        double Yn = (double)Yr[m1 - 1];
        for (int i = m1 - 1; i >= 1; i--) {
            Yn = fma(-Wr[i - 1], Yn, (double)Yr[i - 1]);
            Yr[i - 1] = (T)Yn;
        }
        Yr[nslot] = (T)yout_c0;
    }
}

template <int AMER>
__global__ void __launch_bounds__(64) hadi_pass_b_seq(HadiSweepArgs a, int n) {
    typedef double T;
    static_assert(AMER == 0 || AMER == 1, "explicit (U, lambda_bar) pair only");
    const int lane = threadIdx.x;
    const int nrows = a.L.nrows, rowp = a.L.rowp;
    if ((int)blockIdx.x >= a.n_inst * a.ctiles) return;
    const int inst = blockIdx.x / a.ctiles, tile = blockIdx.x - inst * a.ctiles;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;
    const int col = tile * 64 + lane;
    const bool valid = col < rowp;
    const int colc = valid ? col : rowp - 1;
    T *__restrict__ Yc = reinterpret_cast<T *>(a.Y) + (size_t)inst * a.L.inst_stride + colc;
    T *__restrict__ Uc = reinterpret_cast<T *>(a.U) + (size_t)inst * a.L.inst_stride + colc;
    const double *__restrict__ pb = a.pb + (size_t)inst * a.L.nrows_pad * HADI_PBW;
    // forward: y_k = (rhs_k - L y_{k-1} - L2 y_{k-2}) Q, written over the right-hand side
    double ym1 = 0.0, ym2 = 0.0;
    for (int k0 = 0; k0 < nrows; k0 += 8) {
        double rhs[8];
#pragma unroll
        for (int q = 0; q < 8; q++) rhs[q] = (k0 + q < nrows) ? (double)Yc[(size_t)(k0 + q) * rowp] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (k0 + q < nrows) {
                const double *t = pb + (size_t)(k0 + q) * HADI_PBW;
                const double yk = (rhs[q] - t[PB_L] * ym1 - t[PB_L2] * ym2) * t[PB_Q];
                if (valid) Yc[(size_t)(k0 + q) * rowp] = yk;
                ym2 = ym1;
                ym1 = yk;
            }
        }
    }
    // backward: x_k = y_k - C x_{k+1} - C2 x_{k+2}  (+ Ikonen-Toivanen projection, device_solver.hpp:358-372)
    const double *__restrict__ P0 = (AMER == 1) ? a.U0 + (size_t)inst * a.L.inst_stride + colc : nullptr;
    double *__restrict__ Lc = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride + colc : nullptr;
    const double dt = ip.dt;
    const bool is_smax = (col == a.pos_m1);
    double xp1 = 0.0, xp2 = 0.0;
    for (int k = nrows - 1; k >= 0; k--) {
        const double *t = pb + (size_t)k * HADI_PBW;
        const double yk = Yc[(size_t)k * rowp];
        const double xk = yk - t[PB_C] * xp1 - t[PB_C2] * xp2;
        xp2 = xp1;
        xp1 = xk;
        if constexpr (AMER == 1) {
            const double lamv = Lc[(size_t)k * rowp], pay = P0[(size_t)k * rowp];
            const double un = fmax(xk - dt * lamv, pay);
            double ln = fmax(0.0, lamv + (pay - xk) / dt);
            if (is_smax) ln = 0.0;
            if (valid) {
                Uc[(size_t)k * rowp] = (T)un;
                Lc[(size_t)k * rowp] = ln;
            }
        } else if (valid) {
            Uc[(size_t)k * rowp] = (T)xk;
        }
    }
}

