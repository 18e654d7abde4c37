// hadi_k_small.h -- LDS-resident kernels of the calibration-size grids (hadi_small_kernel, hadi_small_seq_kernel, hadi_small_seq2_kernel).
// Part of libhadi's device code: include through hadi_kernels.h (which fixes the order).
#pragma once

// ------------------------------------------------------------------------------------------------
// Small grids (the reference's calibration / perf-harness sizes, e.g. 50x25: perfomance_test.cpp:46-57):
// the whole instance lives in LDS and ONE launch runs the entire time loop -- no HBM traffic and no kernel
// boundaries inside the loop.  Block = W wavefronts <-> one instance (W = 4: measured faster than 16 on MI355X,
// 0.37 vs 0.46 ms for 500 instances of 50x25x20 -- more blocks per CU beat more waves per instance).  Per step: the row pass is the same
// hadi_row_step as above (rows taken straight from the LDS-resident state, Y written to LDS), then the
// column pass walks each column sequentially in LDS (single chunk: m2+1 <= HADI_LC), with the American
// projection; discrete dividends are applied in place.
struct HadiSmallArgs {
    const int *div_flag;        // dividend index applied at the START of step n (n = 1..Nmax), or -1; nullptr = none.
                                // Instance k reads div_flag[k*flag_stride + n-1]: flag_stride 0 = one shared (N, dt)
    int flag_stride;
    const double *div_amounts;  // device copies of the schedule
    const double *div_pcts;
    const double *vec_s;        // [n_inst][m1+1] (dividend interpolation)
    int Nmax;
    const int *order;           // block b solves instance order[b] (longest time loops first), nullptr = identity
};

template <int B, int W, bool AMER>
__global__ void __launch_bounds__(64 * W) hadi_small_kernel(HadiSweepArgs a, HadiSmallArgs sm) {
    HADI_DYN_SMEM(double, smem);
    constexpr int G = 1, NT = 64 * W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    if ((int)blockIdx.x >= a.n_inst) return;
    // multi-maturity batches: instances with many time steps are dispatched first, short ones fill the tail
    const int inst = sm.order ? sm.order[blockIdx.x] : (int)blockIdx.x;
    const HadiInstPar ip = a.ipar[inst];
    const int nrows = a.L.nrows, rowp = a.L.rowp, m1 = a.L.m1;
    constexpr int c0slot = 64 * B;
    // LDS map: U with two zero rows above and below, Y, [lambda, payoff], coefficients, row table, column table
    const int rows_l = nrows + 4;
    double *Ul = smem + 2 * rowp;                 // row 0 of U (rows -2, -1 and nrows, nrows+1 are zero)
    double *Yl = smem + (size_t)rows_l * rowp;    // nrows rows
    double *LAMl = Yl + (size_t)nrows * rowp;
    double *U0l = LAMl + (AMER ? (size_t)nrows * rowp : 0);
    double *coef = U0l + (AMER ? (size_t)nrows * rowp : 0);
    double *rtab = coef + 4 * 64 * B;
    double *ptab = rtab + (size_t)nrows * HADI_RCL;

    double *__restrict__ Ug = a.U + (size_t)inst * a.L.inst_stride;
    // zero the halo rows of U and ALL of Y: the column pass also sweeps the pad slots of every row, which the row pass
    // never writes -- whatever LDS held there (possibly NaN) would reach U's pad slot, and lane 63 multiplies that slot
    // by a zero coefficient (NaN * 0 = NaN)
    for (int e = tid; e < (rows_l + nrows) * rowp; e += NT) smem[e] = 0.0;
    __syncthreads();
    for (int e = tid; e < nrows * rowp; e += NT) Ul[e] = Ug[e];
    if constexpr (AMER) {
        const double *__restrict__ P0g = a.U0 + (size_t)inst * a.L.inst_stride;
        for (int e = tid; e < nrows * rowp; e += NT) {
            U0l[e] = P0g[e];
            LAMl[e] = 0.0;  // lambda_bar <- 0, device_solver.hpp:310-313
        }
    }
    {
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 64 * B;
        for (int e = tid; e < 4 * 64 * B; e += NT) coef[e] = sc[e];
        const double *__restrict__ rg = a.rowc + (size_t)inst * nrows * HADI_RC;
        for (int e = tid; e < nrows * HADI_RCL; e += NT) rtab[e] = rg[(e / HADI_RCL) * HADI_RC + e % HADI_RCL];
        const double *__restrict__ pg = a.pb + (size_t)inst * a.L.nrows_pad * HADI_PBW;
        for (int e = tid; e < nrows * HADI_PBW; e += NT) ptab[e] = pg[e];
    }
    HadiRowCtx c;
    c.lane = lane; c.half = 0; c.wrow = wave; c.rowp = rowp;
    c.dt = ip.dt; c.thdt = ip.thdt; c.qd = ip.q; c.half_rd = ip.half_rd;
    c.hr0 = ip.hr0; c.inv0 = 1.0 / (1.0 + ip.thdt * ip.hr0);
    c.Yi = Yl; c.Li = AMER ? LAMl : nullptr;
    c.rowc = rtab; c.j0 = 0;
    c.b2r = a.b2row + (size_t)inst * rowp;
    c.coef = coef; c.xch = nullptr; c.R1i = nullptr; c.C2i = nullptr; c.err = a.err; c.debug = 0;
    c.payrow = nullptr; c.inv_dt = 0.0; c.m1_lane = -1; c.m1_r = -1;
    {
        const int ifirst = 1 + B * lane;
        c.posL = hadi_pos(B, G, ifirst - 1);
        c.posR = (ifirst + B <= 64 * B) ? hadi_pos(B, G, ifirst + B) : c0slot + 1;
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c.stamp_acc_ = stamp_store_;
#endif
    __syncthreads();

    const int N = ip.N < sm.Nmax ? ip.N : sm.Nmax;
    HADI_STAMP_DECL(c.stamp_acc_)
    const double *__restrict__ vs = sm.vec_s ? sm.vec_s + (size_t)inst * (m1 + 1) : nullptr;
    for (int n = 1; n <= N; n++) {
        // ---- discrete dividend at the start of the step (device_solver.hpp:448-504) ---------------
        const int dv = sm.div_flag ? sm.div_flag[(size_t)inst * sm.flag_stride + n - 1] : -1;
        if (dv >= 0) {
            for (int e = tid; e < nrows * rowp; e += NT) Yl[e] = Ul[e];  // U_temp
            __syncthreads();
            const double amount = sm.div_amounts[dv], pct = sm.div_pcts[dv];
            for (int e = tid; e < nrows * (m1 + 1); e += NT) {
                const int j = e / (m1 + 1), i = e - j * (m1 + 1);
                const double *src = Yl + (size_t)j * rowp;
                const double new_s = vs[i] * (1.0 - pct) - amount;
                double out = ip.put ? src[c0slot] : 0.0;  // ex-dividend spot <= 0: a call is worth 0 (device_solver.hpp:499-503), a put its s = 0 value
                if (new_s > 0) {
                    int lo = 0, hi = m1 + 1;  // first k with s[k] > new_s (0 if none)
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (vs[mid] > new_s) hi = mid;
                        else lo = mid + 1;
                    }
                    const int idx = (lo <= m1) ? lo : 0;
                    if (idx > 0) {
                        const double s_low = vs[idx - 1], s_high = vs[idx];
                        const double weight = (new_s - s_low) / (s_high - s_low);
                        out = (1.0 - weight) * src[hadi_pos(B, G, idx - 1)] + weight * src[hadi_pos(B, G, idx)];
                    } else {
                        out = src[c0slot];
                    }
                }
                Ul[(size_t)j * rowp + hadi_pos(B, G, i)] = out;
            }
            __syncthreads();
        }
        // ---- row pass: 4 rows at a time, straight out of LDS -----------------------------------------
        c.e_nm1 = exp(ip.bc_rate * ip.dt * (n - 1));
        c.e_n = exp(ip.bc_rate * ip.dt * n);
        HADI_STAMP(8);
        for (int J = 0; J < nrows; J += W) {
            const int j = J + wave;
            if (j < nrows) {
                const double *r0 = Ul + (size_t)j * rowp;
                if (j == nrows - 1)
                    hadi_row_step<B, G, AMER, true>(c, true, j, r0 - 2 * rowp, r0 - rowp, r0, r0 + rowp, r0 + 2 * rowp);
                else
                    hadi_row_step<B, G, AMER, false>(c, true, j, r0 - 2 * rowp, r0 - rowp, r0, r0 + rowp, r0 + 2 * rowp);
            }
        }
        HADI_STAMP(10);  // row pass
        __syncthreads();
        HADI_STAMP(9);  // barrier
        // ---- column pass: one thread per storage column, sequential pentadiagonal sweeps in LDS --------
        for (int col = tid; col < rowp; col += NT) {
            // rounds of eight rows: the independent LDS reads first, then the dependent recurrence (as in hadi_small_seq_kernel)
            double ym1 = 0.0, ym2 = 0.0;
            int k = 0;
            for (; k + 8 <= nrows; k += 8) {
                double yv[8];
#pragma unroll
                for (int q = 0; q < 8; q++) yv[q] = Yl[(size_t)(k + q) * rowp + col];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double *t = ptab + (size_t)(k + q) * HADI_PBW;
                    const double yk = (yv[q] - t[PB_L] * ym1 - t[PB_L2] * ym2) * t[PB_Q];
                    Yl[(size_t)(k + q) * rowp + col] = yk;
                    ym2 = ym1;
                    ym1 = yk;
                }
            }
            for (; k < nrows; k++) {
                const double *t = ptab + (size_t)k * HADI_PBW;
                const double yk = (Yl[(size_t)k * rowp + col] - t[PB_L] * ym1 - t[PB_L2] * ym2) * t[PB_Q];
                Yl[(size_t)k * rowp + col] = yk;
                ym2 = ym1;
                ym1 = yk;
            }
            double xp1 = 0.0, xp2 = 0.0;
            k = nrows - 1;
            if constexpr (!AMER) {
                for (; k >= 7; k -= 8) {
                    double yv[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) yv[q] = Yl[(size_t)(k - q) * rowp + col];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double *t = ptab + (size_t)(k - q) * HADI_PBW;
                        const double xk = yv[q] - t[PB_C] * xp1 - t[PB_C2] * xp2;
                        xp2 = xp1;
                        xp1 = xk;
                        Ul[(size_t)(k - q) * rowp + col] = xk;
                    }
                }
            }
            for (; k >= 0; k--) {
                const double *t = ptab + (size_t)k * HADI_PBW;
                const double xk = Yl[(size_t)k * rowp + col] - t[PB_C] * xp1 - t[PB_C2] * xp2;
                xp2 = xp1;
                xp1 = xk;
                if constexpr (AMER) {  // Ikonen-Toivanen projection, device_solver.hpp:358-372
                    const size_t off = (size_t)k * rowp + col;
                    const double lamv = LAMl[off], pay = U0l[off];
                    Ul[off] = fmax(xk - ip.dt * lamv, pay);
                    double ln = fmax(0.0, lamv + (pay - xk) / ip.dt);
                    if (col == a.pos_m1) ln = 0.0;
                    LAMl[off] = ln;
                } else {
                    Ul[(size_t)k * rowp + col] = xk;
                }
            }
        }
        __syncthreads();
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (lane == 0)
        for (int k = 0; k < 12; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
    for (int e = tid; e < nrows * rowp; e += NT) Ug[e] = Ul[e];
    if constexpr (AMER) {
        double *__restrict__ Lg = a.LAM + (size_t)inst * a.L.inst_stride;
        for (int e = tid; e < nrows * rowp; e += NT) Lg[e] = LAMl[e];
    }
}

// ------------------------------------------------------------------------------------------------
// Small grids, European / dividend sweeps: ONE wavefront per instance, lines solved SEQUENTIALLY, one line per lane.
// The kernel above runs the big-grid row step on 51-node rows: a whole wavefront and six cyclic-reduction levels (54
// ds_bpermute, ~400 instructions) per row -- at one node per lane almost all of it is overhead, and 26 rows x 8
// wavefronts cost ~10 us per time step.  Here the roles are turned round, as in the reference's own team kernels
// (hes_a1_kernels.hpp:139-161, one thread per v-row; hes_a2_shuffled_kernels.hpp:243-299, one thread per s-column):
//   row pass     lane j <-> v-row j (nrows <= 33 lanes busy) walks i = 1 .. m1: explicit operators from a sliding window of
//                three columns (five new LDS values per node), Y0, forward Thomas with the pivot recomputed on the fly; the
//                back substitution walks i = m1 .. 1.  No cross-lane traffic at all.
//   column pass  lane i <-> s-column i: pentadiagonal forward / backward sweep with the precomputed factors.
// State in LDS in NATURAL order, pitch odd (conflict-free both ways): U (two zero halo rows above and below) and Y.  The
// forward sweep needs three values per node for the way back (the normalised right-hand side, the multiplier c', and the
// explicit A2 correction of the output) but only two arrays exist: the output is rewritten as
//   Y_i = x_i + corr_i = (ys_i + corr_i + c'_i corr_{i+1}) - c'_i Y_{i+1} = g_i - c'_i Y_{i+1},
// g_i goes to Y, and c'_i goes to column i-1 of U's own row -- every lane is at the same i (one wavefront), so that column
// has been consumed by all of them.  The column pass rebuilds U completely.  ~45 instructions per node against ~10 x that.
struct HadiSmallSeqLayout {
    int pitch;     // doubles per row in LDS: odd, >= m1 + 3 (columns m1 + 1, m1 + 2 stay zero: the s-neighbour of the last
                   // node and the column the one-ahead fetch touches behind it)
    int off_y, off_coef, off_b2, off_zero, off_dummy, off_ptab, total;  // offsets in doubles: U starts at 0 (nrows rows)
};
HADI_HD inline HadiSmallSeqLayout hadi_small_seq_layout(int m1, int nrows) {
    HadiSmallSeqLayout l;
    l.pitch = (m1 + 3) | 1;
    l.off_y = nrows * l.pitch;
    l.off_coef = l.off_y + nrows * l.pitch;
    l.off_coef = (l.off_coef + 1) & ~1;  // 16-byte aligned quads
    l.off_b2 = l.off_coef + 4 * (m1 + 2);
    l.off_zero = l.off_b2 + (m1 + 2);       // a row of zeros: the "b2 row" of every v-row but the last
    l.off_dummy = l.off_zero + (m1 + 2);    // where the idle lanes (>= nrows) put their results
    l.off_ptab = l.off_dummy + l.pitch;
    l.total = l.off_ptab + nrows * 5;
    return l;
}

template <int B>
__global__ void __launch_bounds__(64) hadi_small_seq_kernel(HadiSweepArgs a, HadiSmallArgs sm) {
    HADI_DYN_SMEM(double, smem);
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= a.n_inst) return;
    const int inst = sm.order ? sm.order[blockIdx.x] : (int)blockIdx.x;
    const HadiInstPar ip = a.ipar[inst];
    const int nrows = a.L.nrows, rowp = a.L.rowp, m1 = a.L.m1;
    const HadiSmallSeqLayout Ls = hadi_small_seq_layout(m1, nrows);
    const int PL = Ls.pitch;
    double *Ul = smem;  // row 0 of U
    double *Yl = smem + Ls.off_y;
    double *coefl = smem + Ls.off_coef;  // [i][4]: Bm, Bp, Dm, Dp of node i
    double *b2l = smem + Ls.off_b2;
    double *ptab = smem + Ls.off_ptab;   // [k][5]: L, L2, Q, C, C2
    double *__restrict__ Ug = a.U + (size_t)inst * a.L.inst_stride;

    for (int e = lane; e < Ls.total; e += 64) smem[e] = 0.0;
    __syncthreads();
    for (int e = lane; e < nrows * (m1 + 1); e += 64) {
        const int j = e / (m1 + 1), i = e - j * (m1 + 1);
        Ul[j * PL + i] = Ug[(size_t)j * rowp + hadi_pos(B, 1, i)];
    }
    {
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 64 * B;
        for (int e = lane; e < 4 * (m1 + 1); e += 64) {
            const int i = e >> 2, k = e & 3;
            coefl[e] = (i >= 1) ? sc[k * 64 * B + hadi_pos(B, 1, i)] : 0.0;
        }
        const double *__restrict__ b2g = a.b2row + (size_t)inst * rowp;
        for (int i = lane; i <= m1; i += 64) b2l[i] = b2g[hadi_pos(B, 1, i)];
        const double *__restrict__ pg = a.pb + (size_t)inst * a.L.nrows_pad * HADI_PBW;
        for (int e = lane; e < nrows * 5; e += 64) ptab[e] = pg[(e / 5) * HADI_PBW + e % 5];
    }
    // this lane's v-row: its table entry stays in registers for the whole time loop
    const int j = lane;
    const bool act = j < nrows;
    const bool last = (j == nrows - 1);
    double v = 0.0, wm = 0.0, wz = 0.0, wp = 0.0, a2l2 = 0.0, a2l1 = 0.0, a2m = 0.0, a2u1 = 0.0, a2u2 = 0.0, b1val = 0.0;
    int b1col = -1;
    bool b1_at0 = false;
    if (act) {
        const double *__restrict__ rc = a.rowc + ((size_t)inst * nrows + j) * HADI_RC;
        v = rc[RC_V]; wm = rc[RC_WM]; wz = rc[RC_WZ]; wp = rc[RC_WP];
        a2l2 = rc[RC_L2]; a2l1 = rc[RC_L1]; a2m = rc[RC_M]; a2u1 = rc[RC_U1]; a2u2 = rc[RC_U2];
        b1val = rc[RC_B1VAL];
        const int b1raw = (int)rc[RC_B1COL];
        b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;  // (two entries on one v-row: m2 > m1 only)
        b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
    }
    const double dt = ip.dt, thdt = ip.thdt, qd = ip.q, half_rd = ip.half_rd;
    const double inv0 = 1.0 / (1.0 + ip.thdt * ip.hr0);
    const double *urow = Ul + (act ? j : 0) * PL;  // (idle lanes walk row 0 and store nothing)
    // The v-neighbours j-2 .. j+2, clamped to the grid instead of zero halo rows (1.7 KB that cost the sixth instance per CU):
    // a clamped row only ever meets a zero weight -- the first / last rows of A0 and A2 have no entries beyond the grid.
    // The v-neighbours j-2 .. j+2 of a column are the SAME column in the neighbouring LANES' rows: one LDS read of the own row
    // and four wave shifts (DPP) instead of five LDS reads per node -- at six wavefronts per CU the one LDS pipe, not the
    // SIMDs, is what this kernel fills (round 3).  Beyond the grid the shifts deliver 0 or an idle lane's (finite) value of
    // row 0; either only ever meets a zero weight -- the first / last rows of A0 and A2 have no entries beyond the grid.
    auto col5 = [&](const double own, double &m2v, double &m1v, double &p1v, double &p2v) {
        m1v = hadi_lane_prev(own); m2v = hadi_lane_prev(m1v);
        p1v = hadi_lane_next(own); p2v = hadi_lane_next(p1v);
    };
    double *yrow = act ? Yl + j * PL : smem + Ls.off_dummy;  // (idle lanes store into a dummy row)
    double *crow = act ? Ul + j * PL : smem + Ls.off_dummy;  // column i - 1 of this row receives c'_i
    const double *b2p = last ? b2l : smem + Ls.off_zero;      // b2 lives on the last v-row only
    __syncthreads();

    const int N = ip.N < sm.Nmax ? ip.N : sm.Nmax;
    const double *__restrict__ vs = sm.vec_s ? sm.vec_s + (size_t)inst * (m1 + 1) : nullptr;
    for (int n = 1; n <= N; n++) {
        // ---- discrete dividend at the start of the step (device_solver.hpp:448-504) ---------------
        const int dv = sm.div_flag ? sm.div_flag[(size_t)inst * sm.flag_stride + n - 1] : -1;
        if (dv >= 0) {
            for (int e = lane; e < nrows * PL; e += 64) Yl[e] = Ul[e];  // U_temp
            __syncthreads();
            const double amount = sm.div_amounts[dv], pct = sm.div_pcts[dv];
            for (int e = lane; e < nrows * (m1 + 1); e += 64) {
                const int jj = e / (m1 + 1), i = e - jj * (m1 + 1);
                const double *src = Yl + jj * PL;
                const double new_s = vs[i] * (1.0 - pct) - amount;
                double out = ip.put ? src[0] : 0.0;  // ex-dividend spot <= 0: a call is worth 0 (device_solver.hpp:499-503), a put its s = 0 value
                if (new_s > 0) {
                    int lo = 0, hi = m1 + 1;  // first k with s[k] > new_s (0 if none)
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (vs[mid] > new_s) hi = mid;
                        else lo = mid + 1;
                    }
                    const int idx = (lo <= m1) ? lo : 0;
                    if (idx > 0) {
                        const double s_low = vs[idx - 1], s_high = vs[idx];
                        const double weight = (new_s - s_low) / (s_high - s_low);
                        out = (1.0 - weight) * src[idx - 1] + weight * src[idx];
                    } else {
                        out = src[0];
                    }
                }
                Ul[jj * PL + i] = out;
            }
            __syncthreads();
        }
        const double e_nm1 = exp(ip.bc_rate * ip.dt * (n - 1));  // device_solver.hpp:238
        const double e_n = exp(ip.bc_rate * ip.dt * n);          // device_solver.hpp:246
        const double cb1 = dt * e_nm1 + thdt * (e_n - e_nm1);
        const double b1l = b1val * cb1;
        // ---- row pass: lane <-> v-row, i = 1 .. m1 (same formulas as hadi_row_step) --------------------------------
        // Only the lanes that own a v-row run the sweeps: the LDS moves 16 or 8 bytes per ACTIVE lane, and with nrows of 64 lanes
        // busy that is less than half of what the idle lanes' dummy walk used to drag through the CU's one LDS pipe.  (The wave
        // shifts deliver 0 from a switched-off lane: the row beyond the last one only ever meets a zero weight.  The emulator's
        // lane threads all have to take part in its collective shuffles, so there every lane still runs.)
#if defined(HADI_EMU)
        const bool rowrun = true;
#else
        const bool rowrun = act;
#endif
        if (rowrun) {
        // column i = 0 (A0 and A1 rows are zero there; only A2 and the boundary act)
        const double c00 = urow[0];
        double c0m2, c0m1, c0p1, c0p2;
        col5(c00, c0m2, c0m1, c0p1, c0p2);
        // first interior column, raw: rows j-2 .. j+2
        double r_0 = urow[1], r_m2, r_m1, r_p1, r_p2;
        col5(r_0, r_m2, r_m1, r_p1, r_p2);
        double yout_c0, x0;
        {
            const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c00 + a2u1 * c0p1 + a2u2 * c0p2;
            const double b1c0 = b1_at0 ? b1val : 0.0;
            const double b2c0 = b2p[0];
            const double a1c0 = -ip.hr0 * c00;  // A1 row 0: empty for the call (hr0 = 0), the reaction term for the put
            double y0c0 = c00 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1);
            y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
            const double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
            x0 = y0c0 * inv0;  // A1 row 0 is decoupled: the identity for the call (hes_a1_kernels.hpp:56-61)
            yout_c0 = x0 + c2c0;
        }
        hadi_wave_rendezvous();  // (emulator: everyone has read column 0 and 1 before c' overwrites column 0)
        double u_prev = c00, u_cur = r_0;
        double t_prev = wm * c0m1 + wz * c00 + wp * c0p1;
        double t_cur = wm * r_m1 + wz * r_0 + wp * r_p1;
        double a2u_cur = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
        double b2c = b2p[1];
        double corr_cur = thdt * (b2c * e_n - (a2u_cur + b2c * e_nm1));
        // raw values of column 2 (column m1 + 1 is the zero spare)
        r_0 = urow[2];
        col5(r_0, r_m2, r_m1, r_p1, r_p2);
        // x_0 is known and moves to the right-hand side of node 1: with ys_0 = x_0 and c'_0 = 0 the general step does exactly
        // that (pivot im - il 0, right-hand side y - il x_0)
        double cp_prev = 0.0, ys_prev = x0;
        // One node of the sweep.  On entry r_* hold the raw column i + 1; `cB`, `cD` are node i's coefficients, `b2n` the b2
        // entry of node i + 1.  The caller refills r_* with column i + 2 afterwards.
        auto node = [&](int i, const double2 cB, const double2 cD, const double b2n) {
            const double u_next = r_0;
            const double t_next = wm * r_m1 + wz * r_0 + wp * r_p1;
            const double a2u_next = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
            const double lo = fma(v, cD.x, qd * cB.x);
            const double up = fma(v, cD.y, qd * cB.y);
            const double mn = -((lo + up) + half_rd);
            const double A1U = lo * u_prev + mn * u_cur + up * u_next;
            const double A0U = cB.x * t_prev - (cB.x + cB.y) * t_cur + cB.y * t_next;
            // Y0 = U + dt (A0U + A1U + A2U + b e_{n-1}) + theta dt (b1 e_n - (A1U + b1 e_{n-1})), device_solver.hpp:236-250
            double S = A0U + A1U + a2u_cur;
            S += b2c * e_nm1;
            double y = fma(dt, S, u_cur);
            y = fma(-thdt, A1U, y);
            y += (i == b1col) ? b1l : 0.0;
            const double il = -thdt * lo;
            const double im = 1.0 - thdt * mn;
            const double iu = -thdt * up;
            const double inv = hadi_rcp(fma(-il, cp_prev, im));
            const double cp = iu * inv;
            const double ys = fma(-il, ys_prev, y) * inv;
            const double corr_next = thdt * (b2n * e_n - (a2u_next + b2n * e_nm1));
            yrow[i] = ys + corr_cur + cp * corr_next;  // g_i  (c'_{m1} = 0: the row ends there)
            crow[i - 1] = cp;
            u_prev = u_cur; u_cur = u_next;
            t_prev = t_cur; t_cur = t_next;
            a2u_cur = a2u_next; corr_cur = corr_next; b2c = b2n;
            cp_prev = cp; ys_prev = ys;
        };
        // Rounds of FOUR nodes: the own-row values of the columns i + 2 .. i + 5, the four nodes' coefficients and b2 entries
        // are all read first -- 16 independent LDS reads behind ONE wait -- then the four dependent steps run without touching
        // the LDS return path.  (Round 2 read five rows' values per node and waited for each node's coefficient and b2 reads:
        // ~900 LDS instructions per time step, and at the six wavefronts per CU that the 26 KB of an instance allow, the CU's
        // one LDS pipe was the busiest unit.)  Column indices reach i + 5 <= m1 + 2: the two zero spare columns of the pitch.
        int i = 1;
        for (; i + 3 <= m1; i += 4) {
            double Rm2[4], Rm1[4], R0[4], Rp1[4], Rp2[4], b2q[4];
            double2 cBq[4], cDq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                R0[q] = urow[i + 2 + q];
                cBq[q] = *reinterpret_cast<const double2 *>(coefl + 4 * (i + q));
                cDq[q] = *reinterpret_cast<const double2 *>(coefl + 4 * (i + q) + 2);
                b2q[q] = b2p[i + 1 + q];  // (entry m1 + 1 is zero)
            }
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");  // (the reads stay in front of the four steps' stores)
#endif
            hadi_wave_rendezvous();  // (emulator: every lane has read its columns before anybody's c' lands in them)
#pragma unroll
            for (int q = 0; q < 4; q++) col5(R0[q], Rm2[q], Rm1[q], Rp1[q], Rp2[q]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                node(i + q, cBq[q], cDq[q], b2q[q]);
                r_m2 = Rm2[q]; r_m1 = Rm1[q]; r_0 = R0[q]; r_p1 = Rp1[q]; r_p2 = Rp2[q];
                hadi_wave_rendezvous();  // (emulator: the lanes walk in lock step on the GPU)
            }
        }
        for (; i <= m1; i++) {  // the last m1 mod 4 nodes, one at a time
            const double n_0 = urow[i + 2];  // (column <= m1 + 2: a zero column)
            double n_m2, n_m1, n_p1, n_p2;
            col5(n_0, n_m2, n_m1, n_p1, n_p2);
            const double2 cB = *reinterpret_cast<const double2 *>(coefl + 4 * i);      // Bm, Bp
            const double2 cD = *reinterpret_cast<const double2 *>(coefl + 4 * i + 2);  // Dm, Dp
            const double b2n = b2p[i + 1];
            hadi_wave_rendezvous();
            node(i, cB, cD, b2n);
            r_m2 = n_m2; r_m1 = n_m1; r_0 = n_0; r_p1 = n_p1; r_p2 = n_p2;
            hadi_wave_rendezvous();  // (emulator: the lanes walk in lock step on the GPU)
        }
        // back substitution on the output itself: Y_i = g_i - c'_i Y_{i+1}
        {   // (idle lanes walk their dummy row)
            double Yn = yrow[m1];
            int i = m1 - 1;
            for (; i >= 8; i -= 8) {  // eight nodes per round: 16 independent LDS reads, then the dependent FMAs
                double g[8], cq[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { g[q] = yrow[i - q]; cq[q] = crow[i - q - 1]; }
#if !defined(HADI_EMU)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    Yn = fma(-cq[q], Yn, g[q]);
                    yrow[i - q] = Yn;
                }
            }
            for (; i >= 1; i--) {
                Yn = fma(-crow[i - 1], Yn, yrow[i]);
                yrow[i] = Yn;
            }
            yrow[0] = yout_c0;
        }
        }  // (rowrun)
        __syncthreads();
        // ---- column pass: lane <-> s-column, sequential pentadiagonal sweeps (hes_a2_shuffled_kernels.hpp:243-299) ----
        // (measured and left out, 50x25 x3000: fetching a node's coefficients one iteration ahead 2.64 -> 2.72 ms; the column
        // held in 33 registers with all loads up front 2.64 -> 2.88 ms)
        for (int col = lane; col <= m1; col += 64) {
            // eight rows per round: the independent LDS reads first, then the dependent recurrence
            double ym1 = 0.0, ym2 = 0.0;
            int k = 0;
            for (; k + 8 <= nrows; k += 8) {
                double yv[8], tL[8], tL2[8], tQ[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {  // (the round's table entries with its column values: 32 reads, one wait)
                    const double *t = ptab + (k + q) * 5;
                    yv[q] = Yl[(k + q) * PL + col];
                    tL[q] = t[PB_L]; tL2[q] = t[PB_L2]; tQ[q] = t[PB_Q];
                }
#if !defined(HADI_EMU)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double yk = (yv[q] - tL[q] * ym1 - tL2[q] * ym2) * tQ[q];
                    Yl[(k + q) * PL + col] = yk;
                    ym2 = ym1;
                    ym1 = yk;
                }
            }
            for (; k < nrows; k++) {
                const double *t = ptab + k * 5;
                const double yk = (Yl[k * PL + col] - t[PB_L] * ym1 - t[PB_L2] * ym2) * t[PB_Q];
                Yl[k * PL + col] = yk;
                ym2 = ym1;
                ym1 = yk;
            }
            double xp1 = 0.0, xp2 = 0.0;
            k = nrows - 1;
            for (; k >= 7; k -= 8) {
                double yv[8], tC[8], tC2[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double *t = ptab + (k - q) * 5;
                    yv[q] = Yl[(k - q) * PL + col];
                    tC[q] = t[PB_C]; tC2[q] = t[PB_C2];
                }
#if !defined(HADI_EMU)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double xk = yv[q] - tC[q] * xp1 - tC2[q] * xp2;
                    xp2 = xp1;
                    xp1 = xk;
                    Ul[(k - q) * PL + col] = xk;
                }
            }
            for (; k >= 0; k--) {
                const double *t = ptab + k * 5;
                const double xk = Yl[k * PL + col] - t[PB_C] * xp1 - t[PB_C2] * xp2;
                xp2 = xp1;
                xp1 = xk;
                Ul[k * PL + col] = xk;
            }
        }
        __syncthreads();
    }
    for (int e = lane; e < nrows * (m1 + 1); e += 64) {
        const int jj = e / (m1 + 1), i = e - jj * (m1 + 1);
        Ug[(size_t)jj * rowp + hadi_pos(B, 1, i)] = Ul[jj * PL + i];
    }
}

// ------------------------------------------------------------------------------------------------
// hadi_small_seq_kernel with TWO instances per wavefront (round 3).  The row sweep above keeps nrows of the 64 lanes busy
// -- 26 for the reference's 50x25 grid -- and is nine tenths of the kernel's instructions: here lanes 0..31 walk the v-rows
// of one instance and lanes 32..63 those of a second one through the SAME instruction stream (per-lane LDS base pointers
// and instance scalars; the wave shifts that fetch the v-neighbours meet zero weights across the boundary between the two
// instances exactly as they do beyond a grid's own first and last row).  The column sweeps (lane <-> s-column) run once
// per instance.  LDS: two instances' arrays per wavefront (52 KB for 50x25: three wavefronts = six instances per CU, as
// before), but a wavefront now retires two instances' time steps in little more than the time of one.  Instances with
// different numbers of time steps (multi-maturity batches, dispatched longest first) simply stop at their own N.
// Needs nrows <= 32.  Same arithmetic, operation by operation, as hadi_small_seq_kernel: the results are bit-identical.
template <int B>
__global__ void __launch_bounds__(64) hadi_small_seq2_kernel(HadiSweepArgs a, HadiSmallArgs sm) {
    HADI_DYN_SMEM(double, smem);
    const int lane = threadIdx.x;
    const int half = lane >> 5, jl = lane & 31;
    const int nrows = a.L.nrows, rowp = a.L.rowp, m1 = a.L.m1;
    const HadiSmallSeqLayout Ls = hadi_small_seq_layout(m1, nrows);
    const int PL = Ls.pitch;
    // the two instances of this wavefront (the second slot of the last block may be empty)
    const int slot0 = 2 * (int)blockIdx.x, slot1 = slot0 + 1;
    if (slot0 >= a.n_inst) return;
    const bool has1 = slot1 < a.n_inst;
    const int inst0 = sm.order ? sm.order[slot0] : slot0;
    const int inst1 = has1 ? (sm.order ? sm.order[slot1] : slot1) : inst0;
    const int inst = half ? inst1 : inst0;             // this lane's instance
    const HadiInstPar ip = a.ipar[inst];               // (per lane: two different structs in the wavefront)
    const HadiInstPar ip0 = a.ipar[inst0], ip1 = a.ipar[inst1];
    const int N0 = ip0.N < sm.Nmax ? ip0.N : sm.Nmax;
    const int N1 = has1 ? (ip1.N < sm.Nmax ? ip1.N : sm.Nmax) : 0;
    const int Nw = N0 > N1 ? N0 : N1;                  // (wave-uniform)
    const int Nl = half ? N1 : N0;                     // this lane's number of time steps
    double *const base0 = smem, *const base1 = smem + Ls.total;
    double *const bl = half ? base1 : base0;           // this lane's instance in LDS
    double *Ul = bl;
    double *Yl = bl + Ls.off_y;
    double *coefl = bl + Ls.off_coef;                  // [i][4]: Bm, Bp, Dm, Dp of node i
    double *b2l = bl + Ls.off_b2;

    for (int e = lane; e < 2 * Ls.total; e += 64) smem[e] = 0.0;
    __syncthreads();
    for (int h = 0; h < (has1 ? 2 : 1); h++) {         // (uniform loops: all 64 lanes copy one instance, then the other)
        const int ih = h ? inst1 : inst0;
        double *bh = h ? base1 : base0;
        const double *__restrict__ Ug = a.U + (size_t)ih * a.L.inst_stride;
        for (int e = lane; e < nrows * (m1 + 1); e += 64) {
            const int j = e / (m1 + 1), i = e - j * (m1 + 1);
            bh[j * PL + i] = Ug[(size_t)j * rowp + hadi_pos(B, 1, i)];
        }
        const double *__restrict__ sc = a.scoef + (size_t)ih * 4 * 64 * B;
        for (int e = lane; e < 4 * (m1 + 1); e += 64) {
            const int i = e >> 2, k = e & 3;
            bh[Ls.off_coef + e] = (i >= 1) ? sc[k * 64 * B + hadi_pos(B, 1, i)] : 0.0;
        }
        const double *__restrict__ b2g = a.b2row + (size_t)ih * rowp;
        for (int i = lane; i <= m1; i += 64) bh[Ls.off_b2 + i] = b2g[hadi_pos(B, 1, i)];
        const double *__restrict__ pg = a.pb + (size_t)ih * a.L.nrows_pad * HADI_PBW;
        for (int e = lane; e < nrows * 5; e += 64) bh[Ls.off_ptab + e] = pg[(e / 5) * HADI_PBW + e % 5];
    }
    // this lane's v-row of its instance: the table entry stays in registers for the whole time loop
    const int j = jl;
    const bool act = j < nrows && (half == 0 || has1);
    const bool last = (j == nrows - 1);
    double v = 0.0, wm = 0.0, wz = 0.0, wp = 0.0, a2l2 = 0.0, a2l1 = 0.0, a2m = 0.0, a2u1 = 0.0, a2u2 = 0.0, b1val = 0.0;
    int b1col = -1;
    bool b1_at0 = false;
    if (act) {
        const double *__restrict__ rc = a.rowc + ((size_t)inst * nrows + j) * HADI_RC;
        v = rc[RC_V]; wm = rc[RC_WM]; wz = rc[RC_WZ]; wp = rc[RC_WP];
        a2l2 = rc[RC_L2]; a2l1 = rc[RC_L1]; a2m = rc[RC_M]; a2u1 = rc[RC_U1]; a2u2 = rc[RC_U2];
        b1val = rc[RC_B1VAL];
        const int b1raw = (int)rc[RC_B1COL];
        b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;
        b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
    }
    const double dt = ip.dt, thdt = ip.thdt, qd = ip.q, half_rd = ip.half_rd;
    const double inv0 = 1.0 / (1.0 + ip.thdt * ip.hr0);
    const double *urow = Ul + (act ? j : 0) * PL;  // (idle lanes walk row 0 and store nothing)
    auto col5 = [&](const double own, double &m2v, double &m1v, double &p1v, double &p2v) {
        m1v = hadi_lane_prev(own); m2v = hadi_lane_prev(m1v);
        p1v = hadi_lane_next(own); p2v = hadi_lane_next(p1v);
    };
    double *const yrow_real = act ? Yl + j * PL : bl + Ls.off_dummy;
    double *const crow_real = act ? Ul + j * PL : bl + Ls.off_dummy;
    const double *b2p = last ? b2l : bl + Ls.off_zero;      // b2 lives on the last v-row only
    __syncthreads();

    for (int n = 1; n <= Nw; n++) {
        // ---- discrete dividends at the start of the step, instance by instance (device_solver.hpp:448-504) ----
        for (int h = 0; h < (has1 ? 2 : 1); h++) {
            const int ih = h ? inst1 : inst0, Nh = h ? N1 : N0;
            const int dv = (sm.div_flag && n <= Nh) ? sm.div_flag[(size_t)ih * sm.flag_stride + n - 1] : -1;
            if (dv >= 0) {  // (wave-uniform)
                double *Uh = h ? base1 : base0, *Yh = Uh + Ls.off_y;
                const double *__restrict__ vs = sm.vec_s + (size_t)ih * (m1 + 1);
                const int put_h = h ? ip1.put : ip0.put;
                for (int e = lane; e < nrows * PL; e += 64) Yh[e] = Uh[e];  // U_temp
                __syncthreads();
                const double amount = sm.div_amounts[dv], pct = sm.div_pcts[dv];
                for (int e = lane; e < nrows * (m1 + 1); e += 64) {
                    const int jj = e / (m1 + 1), i = e - jj * (m1 + 1);
                    const double *src = Yh + jj * PL;
                    const double new_s = vs[i] * (1.0 - pct) - amount;
                    double out = put_h ? src[0] : 0.0;
                    if (new_s > 0) {
                        int lo = 0, hi = m1 + 1;  // first k with s[k] > new_s (0 if none)
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (vs[mid] > new_s) hi = mid;
                            else lo = mid + 1;
                        }
                        const int idx = (lo <= m1) ? lo : 0;
                        if (idx > 0) {
                            const double s_low = vs[idx - 1], s_high = vs[idx];
                            const double weight = (new_s - s_low) / (s_high - s_low);
                            out = (1.0 - weight) * src[idx - 1] + weight * src[idx];
                        } else {
                            out = src[0];
                        }
                    }
                    Uh[jj * PL + i] = out;
                }
                __syncthreads();
            }
        }
        const double e_nm1 = exp(ip.bc_rate * ip.dt * (n - 1));  // device_solver.hpp:238
        const double e_n = exp(ip.bc_rate * ip.dt * n);          // device_solver.hpp:246
        const double cb1 = dt * e_nm1 + thdt * (e_n - e_nm1);
        const double b1l = b1val * cb1;
        // a lane whose instance has finished its own N steps keeps walking (the wave shifts are collective) but stores into the
        // dummy row; on the GPU it is switched off altogether
        const bool live = act && n <= Nl;
        double *const yrow = live ? yrow_real : bl + Ls.off_dummy;
        double *const crow = live ? crow_real : bl + Ls.off_dummy;
#if defined(HADI_EMU)
        const bool rowrun = true;
#else
        const bool rowrun = live;
#endif
        if (rowrun) {
        // ---- row pass: lane <-> v-row of its instance (hadi_small_seq_kernel, operation by operation) ----
        const double c00 = urow[0];
        double c0m2, c0m1, c0p1, c0p2;
        col5(c00, c0m2, c0m1, c0p1, c0p2);
        double r_0 = urow[1], r_m2, r_m1, r_p1, r_p2;
        col5(r_0, r_m2, r_m1, r_p1, r_p2);
        double yout_c0, x0;
        {
            const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c00 + a2u1 * c0p1 + a2u2 * c0p2;
            const double b1c0 = b1_at0 ? b1val : 0.0;
            const double b2c0 = b2p[0];
            const double a1c0 = -ip.hr0 * c00;
            double y0c0 = c00 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1);
            y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
            const double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
            x0 = y0c0 * inv0;
            yout_c0 = x0 + c2c0;
        }
        hadi_wave_rendezvous();
        double u_prev = c00, u_cur = r_0;
        double t_prev = wm * c0m1 + wz * c00 + wp * c0p1;
        double t_cur = wm * r_m1 + wz * r_0 + wp * r_p1;
        double a2u_cur = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
        double b2c = b2p[1];
        double corr_cur = thdt * (b2c * e_n - (a2u_cur + b2c * e_nm1));
        r_0 = urow[2];
        col5(r_0, r_m2, r_m1, r_p1, r_p2);
        double cp_prev = 0.0, ys_prev = x0;
        auto node = [&](int i, const double2 cB, const double2 cD, const double b2n) {
            const double u_next = r_0;
            const double t_next = wm * r_m1 + wz * r_0 + wp * r_p1;
            const double a2u_next = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
            const double lo = fma(v, cD.x, qd * cB.x);
            const double up = fma(v, cD.y, qd * cB.y);
            const double mn = -((lo + up) + half_rd);
            const double A1U = lo * u_prev + mn * u_cur + up * u_next;
            const double A0U = cB.x * t_prev - (cB.x + cB.y) * t_cur + cB.y * t_next;
            double S = A0U + A1U + a2u_cur;
            S += b2c * e_nm1;
            double y = fma(dt, S, u_cur);
            y = fma(-thdt, A1U, y);
            y += (i == b1col) ? b1l : 0.0;
            const double il = -thdt * lo;
            const double im = 1.0 - thdt * mn;
            const double iu = -thdt * up;
            const double inv = hadi_rcp(fma(-il, cp_prev, im));
            const double cp = iu * inv;
            const double ys = fma(-il, ys_prev, y) * inv;
            const double corr_next = thdt * (b2n * e_n - (a2u_next + b2n * e_nm1));
            yrow[i] = ys + corr_cur + cp * corr_next;
            crow[i - 1] = cp;
            u_prev = u_cur; u_cur = u_next;
            t_prev = t_cur; t_cur = t_next;
            a2u_cur = a2u_next; corr_cur = corr_next; b2c = b2n;
            cp_prev = cp; ys_prev = ys;
        };
        int i = 1;
        for (; i + 3 <= m1; i += 4) {
            double Rm2[4], Rm1[4], R0[4], Rp1[4], Rp2[4], b2q[4];
            double2 cBq[4], cDq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                R0[q] = urow[i + 2 + q];
                cBq[q] = *reinterpret_cast<const double2 *>(coefl + 4 * (i + q));
                cDq[q] = *reinterpret_cast<const double2 *>(coefl + 4 * (i + q) + 2);
                b2q[q] = b2p[i + 1 + q];
            }
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");
#endif
            hadi_wave_rendezvous();
#pragma unroll
            for (int q = 0; q < 4; q++) col5(R0[q], Rm2[q], Rm1[q], Rp1[q], Rp2[q]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                node(i + q, cBq[q], cDq[q], b2q[q]);
                r_m2 = Rm2[q]; r_m1 = Rm1[q]; r_0 = R0[q]; r_p1 = Rp1[q]; r_p2 = Rp2[q];
                hadi_wave_rendezvous();
            }
        }
        for (; i <= m1; i++) {
            const double n_0 = urow[i + 2];
            double n_m2, n_m1, n_p1, n_p2;
            col5(n_0, n_m2, n_m1, n_p1, n_p2);
            const double2 cB = *reinterpret_cast<const double2 *>(coefl + 4 * i);
            const double2 cD = *reinterpret_cast<const double2 *>(coefl + 4 * i + 2);
            const double b2n = b2p[i + 1];
            hadi_wave_rendezvous();
            node(i, cB, cD, b2n);
            r_m2 = n_m2; r_m1 = n_m1; r_0 = n_0; r_p1 = n_p1; r_p2 = n_p2;
            hadi_wave_rendezvous();
        }
        {   // back substitution on the output itself: Y_i = g_i - c'_i Y_{i+1}
            double Yn = yrow[m1];
            int ib = m1 - 1;
            for (; ib >= 8; ib -= 8) {
                double g[8], cq[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { g[q] = yrow[ib - q]; cq[q] = crow[ib - q - 1]; }
#if !defined(HADI_EMU)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    Yn = fma(-cq[q], Yn, g[q]);
                    yrow[ib - q] = Yn;
                }
            }
            for (; ib >= 1; ib--) {
                Yn = fma(-crow[ib - 1], Yn, yrow[ib]);
                yrow[ib] = Yn;
            }
            yrow[0] = yout_c0;
        }
        }  // (rowrun)
        __syncthreads();
        // ---- column pass, instance by instance: lane <-> s-column (hes_a2_shuffled_kernels.hpp:243-299) ----
        for (int h = 0; h < (has1 ? 2 : 1); h++) {
            if (n > (h ? N1 : N0)) continue;  // (wave-uniform)
            double *Uh = h ? base1 : base0, *Yh = Uh + Ls.off_y;
            const double *ptab = Uh + Ls.off_ptab;   // [k][5]: L, L2, Q, C, C2
            for (int col = lane; col <= m1; col += 64) {
                double ym1 = 0.0, ym2 = 0.0;
                int k = 0;
                for (; k + 8 <= nrows; k += 8) {
                    double yv[8], tL[8], tL2[8], tQ[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double *t = ptab + (k + q) * 5;
                        yv[q] = Yh[(k + q) * PL + col];
                        tL[q] = t[PB_L]; tL2[q] = t[PB_L2]; tQ[q] = t[PB_Q];
                    }
#if !defined(HADI_EMU)
                    asm volatile("" ::: "memory");
#endif
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double yk = (yv[q] - tL[q] * ym1 - tL2[q] * ym2) * tQ[q];
                        Yh[(k + q) * PL + col] = yk;
                        ym2 = ym1;
                        ym1 = yk;
                    }
                }
                for (; k < nrows; k++) {
                    const double *t = ptab + k * 5;
                    const double yk = (Yh[k * PL + col] - t[PB_L] * ym1 - t[PB_L2] * ym2) * t[PB_Q];
                    Yh[k * PL + col] = yk;
                    ym2 = ym1;
                    ym1 = yk;
                }
                double xp1 = 0.0, xp2 = 0.0;
                k = nrows - 1;
                for (; k >= 7; k -= 8) {
                    double yv[8], tC[8], tC2[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double *t = ptab + (k - q) * 5;
                        yv[q] = Yh[(k - q) * PL + col];
                        tC[q] = t[PB_C]; tC2[q] = t[PB_C2];
                    }
#if !defined(HADI_EMU)
                    asm volatile("" ::: "memory");
#endif
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double xk = yv[q] - tC[q] * xp1 - tC2[q] * xp2;
                        xp2 = xp1;
                        xp1 = xk;
                        Uh[(k - q) * PL + col] = xk;
                    }
                }
                for (; k >= 0; k--) {
                    const double *t = ptab + k * 5;
                    const double xk = Yh[k * PL + col] - t[PB_C] * xp1 - t[PB_C2] * xp2;
                    xp2 = xp1;
                    xp1 = xk;
                    Uh[k * PL + col] = xk;
                }
            }
        }
        __syncthreads();
    }
    for (int h = 0; h < (has1 ? 2 : 1); h++) {
        const int ih = h ? inst1 : inst0;
        const double *bh = h ? base1 : base0;
        double *__restrict__ Ug = a.U + (size_t)ih * a.L.inst_stride;
        for (int e = lane; e < nrows * (m1 + 1); e += 64) {
            const int jj = e / (m1 + 1), i = e - jj * (m1 + 1);
            Ug[(size_t)jj * rowp + hadi_pos(B, 1, i)] = bh[jj * PL + i];
        }
    }
}

