// hadi_k_row_strip.h -- row pass on barrier-free strips (hadi_strip_step, hadi_pass_a_strip: one or two wavefronts per v-row; hadi_pass_a_pairs: two strips per wavefront).
// Part of libhadi's device code: include through hadi_kernels.h (which fixes the order).
#pragma once

// ------------------------------------------------------------------------------------------------
// Strip row pass (8 nodes per lane, one wavefront per v-row: 256 < m1 <= 512).  Same arithmetic as hadi_row_step,
// different data movement: every wavefront owns a strip of RS consecutive v-rows and walks down it ALONE -- no
// barrier in the loop.  The rows j-2, j-1, j of the stencil stay in registers from the previous steps, rows j+1 and
// j+2 sit in the wavefront's private 4-slot LDS ring, rows j+3 and j+4 are in flight (LDS-DMA issued by this
// wavefront, retired by its own counted vmcnt wait).  The s-neighbours of a lane's block come from the adjacent
// lanes (ds_bpermute), the i = 0 column from the row's extra slot, the row's table entry through the scalar cache
// into SGPRs.  Against the shared-ring kernel:
// no block-wide barrier (its wait was ~40 % of a wavefront's time there), twice the rows in flight per CU, a quarter
// of the LDS reads; the price is 4 halo rows per strip read again (mostly L2 hits).
// American P representation on strips: from this many nodes per lane on, the raw P of row j is read again from its ring slot
// (kept one step longer) instead of being carried in registers.  At 8 nodes per lane that ends the spilling (256 VGPRs + 12
// spilled -> 228; 512x256 x256 American: row pass 0.140 -> 0.121 ms per launch); at 4 the kernel fits either way and the
// shorter prefetch costs more than the registers gain (256x128 x512 American puts: 0.0841 -> 0.0866).
#ifndef HADI_STRIP_CREG_MAX_B
#define HADI_STRIP_CREG_MAX_B 4  // strips of at most this many nodes per lane keep the s-coefficients in registers
#endif
#ifndef HADI_AMP_KEEP_MIN_B
#define HADI_AMP_KEEP_MIN_B 8
#endif
template <class T>
struct HadiStripCtxT {
    const double *coef;  // LDS: Bm, Bp, Dm, Dp, each 64*B doubles in row layout
    T *Yi;               // instance base of Y
    double *R1i, *C2i;   // instance bases of the Craig-Sneyd carry-over arrays (MODE 1 writes, MODE 2 reads)
    const double *Li;    // instance base of lambda_bar (American)
    const double *b2r;   // instance b2 row (global)
    int lane, rowp;
    double dt, thdt, e_nm1, e_n;
    double c1, kap;           // 1 + theta dt r_d / 2, (1 - theta) / theta
    double hr0, inv0;         // i = 0 row of A1: reaction term (0 for the call) and 1 / (1 + theta dt hr0)
    double inv_dt;            // P representation: 1 / dt and the (lane, slot) of the s_max node
    int m1_lane, m1_r;
    int half;                 // G = 2: which half of the row this wavefront owns (0: nodes 1..64B, 1: the rest)
    double *xch;              // G = 2: LDS exchange of the wavefront pair, [2 row parities][4 values + 2 tokens + 2 spare]
    int *err;                 // HadiSweepArgs.err / .debug (G = 2 rendezvous)
    int debug;
    HADI_STAMP_ACC
};

// Pair rendezvous flags in LDS (G = 2 strips and the shared ring): release store / acquire load at workgroup scope.
HADI_DEV HADI_FORCEINLINE void hadi_flag_store(int *f, int v) {
#if defined(HADI_EMU)
    __atomic_store_n(f, v, __ATOMIC_RELEASE);
#else
    __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
HADI_DEV HADI_FORCEINLINE int hadi_flag_load(int *f) {
#if defined(HADI_EMU)
    return __atomic_load_n(f, __ATOMIC_ACQUIRE);
#else
    return __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

// Craig-Sneyd corrector on strips: the R1 and C2 rows of a step are REGISTER loads -- ordinary loads of the compiler, on
// purpose.  hipcc retires them with `s_waitcnt vmcnt(k)`, k = the younger vector-memory operations IT knows of (the row stores
// of the step in between); the LDS-DMA prefetch is inline asm and not among them, so that wait also retires the DMA batch
// issued between the loads and the stores: the corrector runs with ONE row of DMA prefetch in flight behind the landed ones
// where the Douglas strips have two.  The first version of round 4 issued these loads from inline asm and retired them by the
// kernel's own counted wait to keep that second row in flight -- and was wrong: registers with a load in flight are values the
// compiler believes ready, and it moved copies of them (the operand set-up of the very asm that was to consume them) ABOVE
// the wait.  Small tests passed (the data had landed by then); 256 instances of 512x256 gave garbage against the build whose
// waits drain everything (tests/test_gpu_parity.py::test_craig_sneyd_on_strips_under_load_equals_full_drains, added first,
// failing).  In-flight registers cannot be handed to a compiler that may copy them.
HADI_DEV HADI_FORCEINLINE double hadi_nt_load(const double *p) {
#if defined(HADI_EMU)
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}
template <int B>
struct HadiCsRow { double r1[B], c2[B], r1c0, c2c0; };  // MODE 2: R1 and C2 of the step's row (block and i = 0 column)
// the row's loads (non-temporal: hadi_get_block_nt); returns the number of vector-memory instructions (2 (B / 2 + 1))
template <int B, int G>
HADI_DEV HADI_FORCEINLINE int hadi_cs_row_load(const double *r1row, const double *c2row, int half, int lane, HadiCsRow<B> &o) {
    hadi_get_block_nt<B, G>(r1row, half, lane, o.r1);
    hadi_get_block_nt<B, G>(c2row, half, lane, o.c2);
#if defined(HADI_EMU)
    o.r1c0 = r1row[64 * B * G];
    o.c2c0 = c2row[64 * B * G];
#else
    o.r1c0 = __builtin_nontemporal_load(r1row + 64 * B * G);
    o.c2c0 = __builtin_nontemporal_load(c2row + 64 * B * G);
#endif
    return 2 * (B / 2 + 1);
}

// MODE 0: Douglas step.  MODE 1 / 2: predictor / corrector of Craig-Sneyd exactly as in hadi_row_step (solver.hpp:781-907):
// MODE 1 is a Douglas row step that also stores R1 = Y1rhs - dt/2 A0U and C2; MODE 2 takes its rows from Y2, forms
// R1 + dt/2 A0 Y2, runs the same A1 solve and adds C2 (cs: the two rows, loaded by the caller).  European sweeps only.
// AMER: 0 European, 1 American with the explicit (U, lambda_bar) pair (lambda_bar loaded here), 2 American in the P
// representation: the caller rebuilt U = max(P, U_0) on the five rows and hands over the raw P of row j (p_raw) and
// lambda_bar of the i = 0 column; lambda_bar = max(0, (U_0 - P)/dt) = (U - P)/dt is formed here, right before the sweep
// that consumes it (formed by the caller it stayed live across the explicit operators and the kernel spilled).
// G = 2 (512 < m1 <= 1024, European): the row is shared by a PAIR of wavefronts, each owning one half.  eb / e0 / ea are
// the values, on the rows behind / at / ahead of j, of the one node next to this half that belongs to the partner; the
// tridiagonal system is split at the boundary exactly as in hadi_row_step (second right-hand side through the cyclic
// reduction, 2x2 system exchanged through LDS), with a rendezvous of the two wavefronts only.
// RSTAB (paired strips, round 4): the second right-hand side of a pair's half-row system -- its coupling column, carried through
// the six levels of the cyclic reduction next to the real right-hand side -- depends on the MATRIX only: on the instance, the
// v-row and the half, not on the time step.  It is computed once per solve (MODE 3: this step with nothing but that column;
// the result, one double per lane, goes to rs_tab) and every later step takes it from there (rs_in, loaded by the caller a
// step ahead): 18 cross-lane fetches and ~25 operations per row and wavefront less in the reduction (the 1024x512 row pass
// -6 % with the updates compiled out; profiles/r04_pair_spike_ab.txt).
template <int B, int AMER, bool LAST, class T = double, int G = 1, int CREG = 0, int MODE = 0, bool RSTAB = false>
HADI_DEV HADI_FORCEINLINE void hadi_strip_step(const HadiStripCtxT<T> &c, int j, const double (&rt)[HADI_RCL],
                                               const double (&um2)[B], const double (&um1)[B], const double (&u0)[B],
                                               const double (&up1)[B], const double (&up2)[B], double c0m2, double c0m1,
                                               double c00, double c0p1, double c0p2, const double (&p_raw)[B],
                                               double lamc0_in, const T *next_row, double (&u_next)[B],
                                               double eb = 0.0, double e0 = 0.0, double ea = 0.0, const T *raw_row = nullptr,
                                               const double *pay_row = nullptr, const double *cf = nullptr,
                                               const HadiCsRow<B> *cs = nullptr, double rs_in = 0.0, double *rs_out = nullptr) {
    static_assert(G == 1 || (G == 2 && (AMER == 0 || sizeof(T) == 8)), "paired strips: American sweeps with the fp64 state only");
    static_assert(MODE == 0 || (AMER == 0 && sizeof(T) == 8), "Craig-Sneyd: European sweeps, fp64 state");
    static_assert(MODE != 3 || (G == 2 && !RSTAB), "MODE 3 builds the coupling-column table of the paired strips");
    static_assert(!RSTAB || G == 2, "the coupling column exists on paired strips only");
    const int lane = c.lane, rowp = c.rowp;
    const int half = (G > 1) ? c.half : 0;
    // P representation: where the raw P of row j comes from -- its ring slot again (8 nodes per lane, one wavefront per row:
    // the kernel keeps that slot one step longer) or the caller's registers (the 3-slot ring of the paired strips has no
    // slot to spare)
    constexpr bool RAW_FROM_RING = (B >= HADI_AMP_KEEP_MIN_B && G == 1);
    // P representation on paired strips: u0 ARRIVES RAW (P of row j) and U = max(P, U_0) is formed where it is used, from
    // the payoff row in LDS (pay_row) -- once for the explicit operators, once more pair by pair inside the forward sweep,
    // where lambda_bar = (U - P)/dt falls out of the same two values.  Carrying U and P (or lambda_bar) side by side is
    // 16 registers more than the 256 this kernel has (hipcc spilled 8 of them into the row loop).
    constexpr bool RAW_U0 = (AMER == 2 && !RAW_FROM_RING && G == 2);
    const bool first_half = (half == 0), last_half = (half == G - 1);
    constexpr int c0slot = 64 * B * G;
    constexpr int NB = B - 1;
    HADI_STAMP_DECL(c.stamp_acc_)
    const double dt = c.dt, thdt = c.thdt, c1 = c.c1, kap = c.kap, e_nm1 = c.e_nm1, e_n = c.e_n;
    // rt = the entries RC_L2 .. RC_WPS of the row's table entry (HADI_SRC0): wm, wz, wp are the SCALED A0 v-weights,
    // -w / (theta dt (r_d - r_f)), to go with the scaled s-coefficient arrays (below)
    const double vth = rt[RC_VTH - HADI_SRC0];
    const double wm = rt[RC_WMS - HADI_SRC0], wz = rt[RC_WZS - HADI_SRC0], wp = rt[RC_WPS - HADI_SRC0];
    const double a2l2 = rt[RC_L2 - HADI_SRC0], a2l1 = rt[RC_L1 - HADI_SRC0], a2m = rt[RC_M - HADI_SRC0], a2u1 = rt[RC_U1 - HADI_SRC0],
                 a2u2 = rt[RC_U2 - HADI_SRC0];
    const double b1val = rt[RC_B1VAL - HADI_SRC0];
    const int b1raw = (int)rt[RC_B1COL - HADI_SRC0];
    const bool b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;  // (two entries on one v-row: m2 > m1 only)
    const int b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
    const int b1e = b1col - 1;
    const int b1half = (G > 1 && b1col >= 1) ? b1e / (64 * B) : 0;  // which wavefront of the pair holds the b1 node
    const int b1el = b1e - b1half * 64 * B;
    const int b1lane = (b1col >= 1 && b1half == half) ? b1el / B : -1;
    const int b1r = b1el - (b1el / B) * B;

    // ---- column i = 0 (A0 and A1 rows are zero there; only A2 and the boundary act) ----------------
    const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c00 + a2u1 * c0p1 + a2u2 * c0p2;
    const double b1c0 = b1_at0 ? b1val : 0.0;
    const double b2c0 = LAST ? c.b2r[c0slot] : 0.0;
    const double lamc0 = (AMER == 1) ? c.Li[(size_t)j * rowp + c0slot] : (AMER == 2) ? lamc0_in : 0.0;
    const double a1c0 = -c.hr0 * c00;  // A1 row 0: empty for the call (hr0 = 0), the reaction term for the put
    double y0c0 = c00 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1 + lamc0);
    y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
    double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
    if constexpr (MODE == 1) {  // A0 is zero on i = 0: R1 = Y1rhs there
        if (lane == 0 && first_half) {
            c.R1i[(size_t)j * rowp + c0slot] = y0c0;
            c.C2i[(size_t)j * rowp + c0slot] = c2c0;
        }
    }
    if constexpr (MODE == 2) {
        y0c0 = cs->r1c0;
        c2c0 = cs->c2c0;
    }
    const double x0 = y0c0 * c.inv0;  // A1 row 0 is decoupled: the identity for the call (hes_a1_kernels.hpp:56-61)
    const double yout_c0 = x0 + c2c0;

    // ---- explicit operators (same evaluation order as hadi_row_step) -------------------------------------
    double tt[B], A2U[B];
    double u0_first = u0[0], u0_last = u0[B - 1];
    if constexpr (RAW_U0) {
        // (the payoff pair by pair, behind a compiler barrier: as one 16-register block it stayed live from the caller's
        // max operations to the end of this loop and the kernel spilled 16 registers into the row loop)
        double payx[B];
#pragma unroll
        for (int r = 0; r < B; r++) {
            if ((r & 1) == 0) {
#if !defined(HADI_EMU)
                asm volatile("" ::: "memory");
#endif
                const double2 pp = *reinterpret_cast<const double2 *>(pay_row + (r >> 1) * 128 * G + 128 * half + 2 * lane);
                payx[r] = pp.x; payx[r + 1 < B ? r + 1 : r] = pp.y;
            }
            const double um = fmax(u0[r], payx[r]);
            tt[r] = wm * um1[r] + wz * um + wp * up1[r];
            A2U[r] = a2l1 * um1[r] + a2m * um + a2u1 * up1[r];
            if (r == 0) u0_first = um;
            if (r == B - 1) u0_last = um;
        }
    } else {
#pragma unroll
        for (int r = 0; r < B; r++) {
            tt[r] = wm * um1[r] + wz * u0[r] + wp * up1[r];
            A2U[r] = a2l1 * um1[r] + a2m * u0[r] + a2u1 * up1[r];
        }
    }
    // The second neighbours j-2, j+2 enter A2 only on the upwind rows (v_j > 1: hes_a2_shuffled_kernels.hpp:131-140) and on
    // row 0 (the gamma stencil): three rows in four have both weights zero -- a wave-uniform branch (the weights sit in
    // SGPRs) around the 2 B FMAs.  fma(0, u, A) = A: bit-identical.
    if (a2l2 != 0.0 || a2u2 != 0.0) {
#pragma unroll
        for (int r = 0; r < B; r++) A2U[r] = fma(a2l2, um2[r], A2U[r]);
#pragma unroll
        for (int r = 0; r < B; r++) A2U[r] = fma(a2u2, up2[r], A2U[r]);
    }
    // s-neighbours of the block: last node of lane-1, first node of lane+1; lane 0 borders i = 0, lane 63 the pad (0)
    double u0L = hadi_lane_prev(u0_last), tL = hadi_lane_prev(tt[B - 1]);
    double u0R = hadi_lane_next(u0_first), tR = hadi_lane_next(tt[0]);
    if (lane == 0 && first_half) {
        u0L = c00;
        tL = wm * c0m1 + wz * c00 + wp * c0p1;
    }
    if constexpr (G > 1) {  // the node across the pair boundary
        const double te = wm * eb + wz * e0 + wp * ea;
        if (lane == 0 && !first_half) { u0L = e0; tL = te; }
        if (lane == 63 && !last_half) { u0R = e0; tR = te; }
    }
    // (lane 63 of the last half: hadi_lane_next delivered the pad's zeros already)
    const double cb1 = dt * e_nm1 + thdt * (e_n - e_nm1);
    const double b1l = (lane == b1lane) ? b1val * cb1 : 0.0;  // this row's b1 entry, in the lane that owns its node

    double lam[B], b2v[B];
    double r1v[B], c2v[B];  // (MODE 1)
    if constexpr (AMER == 1) hadi_get_block<B, G>(c.Li + (size_t)j * rowp, half, lane, lam);
    if constexpr (AMER == 2) {
        // 8 nodes per lane: the raw P of row j is read again from its ring slot, which stays intact through this step (the
        // kernel keeps one slot behind the prefetch for it) -- carried in 16 more registers the kernel spilled
        double praw8[B];
        if constexpr (RAW_FROM_RING) hadi_get_block<B, 1, T>(raw_row, 0, lane, praw8);
        else {
#pragma unroll
            for (int r = 0; r < B; r++) praw8[r] = 0.0;
        }
#pragma unroll
        for (int r = 0; r < B; r++) {
            if constexpr (RAW_U0) lam[r] = 0.0;  // (formed inside the sweep)
            else {
                lam[r] = (u0[r] - (RAW_FROM_RING ? praw8[r] : p_raw[r])) * c.inv_dt;  // exactly max(0, (U_0 - P) / dt): U = max(P, U_0)
                if (lane == c.m1_lane && r == c.m1_r) lam[r] = 0.0;  // s_max keeps lambda_bar = 0, as in hadi_row_step
            }
        }
    }
    if constexpr (LAST && MODE != 2) hadi_get_block<B, G>(c.b2r, half, lane, b2v);

    // ---- Y0 (device_solver.hpp:236-250) fused with the forward sweep of the in-lane Thomas ----------
    // The s-coefficients are read pair by pair inside the sweep (a compiler barrier keeps hipcc from hoisting all 16
    // LDS reads to the top: 32 live doubles there are what pushed this kernel into scratch).
    hadi_set_prio(1);
    double Bm[B], Bp[B], Dm[B], Dp[B];
    double ys[B], ps[B], gs[B], iu[B], cp[B];
    double il_last = 0.0, im_last = 1.0, d_last = 0.0;
    double um[B];  // RAW_U0: U = max(P, U_0) of this row, two nodes ahead of the sweep
    if constexpr (RAW_U0) {
        const double2 pp = *reinterpret_cast<const double2 *>(pay_row + 128 * half + 2 * lane);
        um[0] = fmax(u0[0], pp.x);
        um[1] = fmax(u0[1], pp.y);
    }
#pragma unroll
    for (int r = 0; r < B; r++) {
        if ((r & 1) == 0) {
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");
#endif
            const int q = r >> 1;
            const int co = q * 128 * G + 128 * half + 2 * lane;
            if constexpr (RAW_U0) {
                if (r + 2 < B) {
                    const double2 pp = *reinterpret_cast<const double2 *>(pay_row + co + 128 * G);
                    um[r + 2 < B ? r + 2 : 0] = fmax(u0[r + 2 < B ? r + 2 : 0], pp.x);
                    um[r + 3 < B ? r + 3 : 0] = fmax(u0[r + 3 < B ? r + 3 : 0], pp.y);
                }
            }
            if constexpr (CREG >= 2) {  // (8 nodes per lane: the two diffusion arrays, or only the second, from registers)
                const double2 t0 = *reinterpret_cast<const double2 *>(c.coef + 0 * 64 * B * G + co);
                const double2 t1 = *reinterpret_cast<const double2 *>(c.coef + 1 * 64 * B * G + co);
                Bm[r] = t0.x; Bm[r + 1] = t0.y;
                Bp[r] = t1.x; Bp[r + 1] = t1.y;
                if constexpr (CREG == 2) { Dm[r] = cf[2 * B + r]; Dm[r + 1] = cf[2 * B + r + 1]; }
                else { const double2 t2 = *reinterpret_cast<const double2 *>(c.coef + 2 * 64 * B * G + co); Dm[r] = t2.x; Dm[r + 1] = t2.y; }
                Dp[r] = cf[3 * B + r]; Dp[r + 1] = cf[3 * B + r + 1];
            } else if constexpr (CREG) {
                // 2 and 4 nodes per lane: the lane's s-coefficients stay in registers for the whole strip (cf: [4][B], the
                // kernel has the registers to spare) -- the LDS pipe, shared by all wavefronts of the CU, is what these row
                // widths run out of first (hadi_pass_a_strip)
                Bm[r] = cf[0 * B + r]; Bm[r + 1] = cf[0 * B + r + 1];
                Bp[r] = cf[1 * B + r]; Bp[r + 1] = cf[1 * B + r + 1];
                Dm[r] = cf[2 * B + r]; Dm[r + 1] = cf[2 * B + r + 1];
                Dp[r] = cf[3 * B + r]; Dp[r + 1] = cf[3 * B + r + 1];
            } else {
                const double2 t0 = *reinterpret_cast<const double2 *>(c.coef + 0 * 64 * B * G + co);
                const double2 t1 = *reinterpret_cast<const double2 *>(c.coef + 1 * 64 * B * G + co);
                const double2 t2 = *reinterpret_cast<const double2 *>(c.coef + 2 * 64 * B * G + co);
                const double2 t3 = *reinterpret_cast<const double2 *>(c.coef + 3 * 64 * B * G + co);
                Bm[r] = t0.x; Bm[r + 1] = t0.y;
                Bp[r] = t1.x; Bp[r + 1] = t1.y;
                Dm[r] = t2.x; Dm[r + 1] = t2.y;
                Dp[r] = t3.x; Dp[r + 1] = t3.y;
            }
        }
        const double u0r = RAW_U0 ? um[r] : u0[r];
        const double uL = (r == 0) ? u0L : (RAW_U0 ? um[r == 0 ? 0 : r - 1] : u0[r == 0 ? 0 : r - 1]);
        const double uR = (r == B - 1) ? u0R : (RAW_U0 ? um[r == B - 1 ? r : r + 1] : u0[r == B - 1 ? r : r + 1]);
        const double tl = (r == 0) ? tL : tt[r == 0 ? 0 : r - 1];
        const double tr = (r == B - 1) ? tR : tt[r == B - 1 ? r : r + 1];
        if constexpr (RAW_U0) {
            lam[r] = (u0r - u0[r]) * c.inv_dt;  // exactly max(0, (U_0 - P) / dt)
            if (lane == c.m1_lane && r == c.m1_r) lam[r] = 0.0;
        }
        // I - theta dt A1 directly (theta dt v comes with the row's table entry): il, im, iu; theta dt A1 U from the
        // same three; Y0 - theta dt A1 U = U + dt (A0 U + A2 U + ...) + (1 - theta)/theta (theta dt A1 U).
        // Bm, Bp hold E = -theta dt (r_d - r_f) s beta_s (scaled while the block copied the arrays to LDS): the convection
        // part of il / iu costs no multiplication, and A0 U = s beta_s (x) [w u] = E (x) [(-w / (theta dt (r_d - r_f))) u]
        // comes out of the same arrays with the scaled v-weights of the row table -- 2 operations per node fewer.
        double il = fma(-vth, Dm[r], Bm[r]);
        iu[r] = fma(-vth, Dp[r], Bp[r]);
        const double sm = il + iu[r];
        const double im = c1 - sm;  // 1 + theta dt (lo + up + r_d / 2)
        // theta dt A1 U = -il uL - iu uR + (1 - im) u0   (c1 - c2 = 1)
        const double T1 = fma(-iu[r], uR, fma(-il, uL, fma(-im, u0r, u0r)));
        const double A0U = Bm[r] * tl - (Bm[r] + Bp[r]) * tt[r] + Bp[r] * tr;
        double y;
        if constexpr (MODE == 2) {
            y = fma(0.5 * dt, A0U, cs->r1[r]);  // A0U is A0 applied to Y2 here
        } else {
            double S = A0U + A2U[r];
            if constexpr (LAST) S += b2v[r] * e_nm1;
            if constexpr (AMER) S += lam[r];
            y = fma(dt, S, u0r);
            y = fma(kap, T1, y);
            y = fma(b1l, (r == b1r) ? 1.0 : 0.0, y);  // wave-uniform selector: one FMA with a scalar operand (a scalar branch
                                                      // around a single add measured slower: 0.1108 vs 0.1099 ms per launch)
            if constexpr (MODE == 1) r1v[r] = fma(-0.5 * dt, A0U, y);
        }
        if (r == 0 && lane == 0 && first_half) {  // x_0 is known: move it to the right-hand side
            y -= il * x0;
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
    HADI_STAMPC(26);  // explicit operators + Y0 + forward Thomas
    // reduced (interface) row of this lane
    double ra, rb, rcc, rf, rs = 0.0;
    const bool edge_hi = (G > 1) && !last_half && lane == 63;  // next node belongs to the partner wavefront
    const bool edge_lo = (G > 1) && !first_half && lane == 0;  // previous node belongs to the partner wavefront
    {
        gs[NB - 1] = cp[NB - 1];
#pragma unroll
        for (int r = NB - 2; r >= 0; r--) {
            ys[r] = fma(-cp[r], ys[r + 1], ys[r]);
            ps[r] = fma(-cp[r], ps[r + 1], ps[r]);
            gs[r] = -cp[r] * gs[r + 1];
        }
        double p0n = hadi_lane_next(ps[0]), g0n = hadi_lane_next(gs[0]), y0n = hadi_lane_next(ys[0]);
        if constexpr (G > 1) {
            if (edge_hi) { p0n = 0.0; g0n = 0.0; y0n = 0.0; }
        }
        ra = -il_last * ps[NB - 1];
        rb = im_last - il_last * gs[NB - 1] - iu[B - 1] * p0n;
        rcc = -iu[B - 1] * g0n;
        rf = d_last - il_last * ys[NB - 1] - iu[B - 1] * y0n;
        if constexpr (G > 1) {
            if (edge_hi) { rs = iu[B - 1]; rcc = 0.0; }  // couples to t = first node of the partner's half
            if (edge_lo) { rs = ra; ra = 0.0; }          // couples to the last node of the partner's half
            // (RSTAB: rs is dead from here to the end of the reduction -- the stored image replaces it there)
        }
    }
    HADI_STAMPC(27);  // backward Thomas + reduced row
    // ---- parallel cyclic reduction over the 64 interface unknowns (normalised rows, see hadi_row_step) ----
    {
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
            if constexpr (G > 1 && !RSTAB) {  // the second right-hand side (coupling to the partner's boundary node)
                const double sL = (s == 1) ? hadi_lane_prev(rs) : hadi_lane_get(rs, up_lane);
                const double sR = (s == 1) ? hadi_lane_next(rs) : (s == 32) ? sL : hadi_lane_get(rs, dn_lane);
                rs = fma(-rcc, sR, fma(-ra, sL, rs)) * rn;
            }
            if (s < 32) {
                const double an = -(ra * aL) * rn;
                const double cn = -(rcc * cR) * rn;
                ra = an;
                rcc = cn;
            }
        }
    }
    HADI_STAMPC(28);  // PCR
    hadi_set_prio(0);
    if constexpr (MODE == 3) {  // the table entry of this (row, half, lane); no exchange, no result
        *rs_out = rs;
#pragma unroll
        for (int r = 0; r < B; r++) u_next[r] = 0.0;
        return;
    }
    if constexpr (RSTAB) rs = rs_in;
    // the next row (this step's "row ahead") again from its ring slot, intact until the next step: issued here so that the
    // read flies during the final combination and the stores instead of being waited for at the end of the step
    hadi_get_block<B, G, T>(next_row, half, lane, u_next);
    double X = rf, XL;
    if constexpr (G > 1) {
        // X(l) = rf - bv rs with bv the partner's boundary node.  Publish what the 2x2 system needs (hadi_row_step):
        //   low half, lane 63:  x_hi = A - t Bc   (A = rf, Bc = rs; x_hi = its own X)
        //   high half, lane 0:  t = C - x_hi D    (t = its first node = ys0 - XL ps0 - X gs0, XL = x_hi)
        // into the buffer of this row's parity, then the token behind the values (same lane: the LDS unit sees data before
        // flag).  The partner walks the same strip in the same direction, so it always arrives; it can be at most one row
        // away, hence two buffers are enough.  The poll is bounded (a logic error must not hang the GPU); running out of
        // polls is reported through the handle's error word (hadi_report) and fails the call.
        double *xb = c.xch + 8 * (j & 1);
        int *flags = reinterpret_cast<int *>(xb + 4);
        const int token = j + 1;
        const bool withhold = (c.debug & HADI_DEBUG_WITHHOLD_TOKEN) && half == 1 && j == 1;  // (test hook)
        if (edge_hi) {
            xb[0] = rf;
            xb[1] = rs;
            hadi_flag_store(flags + 0, token);
        }
        if (edge_lo) {
            xb[2] = ys[0] - rf * gs[0];
            xb[3] = ps[0] - rs * gs[0];
            if (!withhold) hadi_flag_store(flags + 1, token);
        }
        hadi_wave_rendezvous();  // (emulator: this wavefront's own publisher lane has written)
        int guard = 0;
        const int polls = HADI_RENDEZVOUS_POLLS(c.debug);
        while (hadi_flag_load(flags + (1 - half)) != token && ++guard < polls) {
#if defined(HADI_EMU)
            sched_yield();
#else
            __builtin_amdgcn_s_sleep(1);
#endif
        }
        if (guard >= polls && lane == 0) hadi_report(c.err, HADI_DEVERR_RENDEZVOUS);
        const double A = xb[0], Bc = xb[1], Cc = xb[2], Dd = xb[3];
        const double xhi = (A - Bc * Cc) * hadi_rcp(1.0 - Bc * Dd);  // last node of the low half (reciprocal + Newton step as in the line solves: the IEEE division sequence is 12 dependent instructions on the pair's critical path)
        const double tlo = Cc - Dd * xhi;                    // first node of the high half
        X = rf - (first_half ? tlo : xhi) * rs;
        XL = hadi_lane_prev(X);
        if (lane == 0 && !first_half) XL = xhi;
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
        if constexpr (MODE == 2) corr = cs->c2[r];
        else if constexpr (LAST) corr = thdt * (b2v[r] * e_n - (A2U[r] + b2v[r] * e_nm1));
        else corr = -thdt * A2U[r];
        yo[r] = x + corr;
        if constexpr (MODE == 1) c2v[r] = corr;
    }
    if constexpr (MODE == 1) {  // (two more row stores per step: the kernel's counted waits add them)
        hadi_put_block_nt<B, G>(c.R1i + (size_t)j * rowp, half, lane, r1v);
        hadi_put_block_nt<B, G>(c.C2i + (size_t)j * rowp, half, lane, c2v);
    }
    // Plain global stores on purpose.  Raw BUFFER stores here (SGPR row offset, one 32-bit lane offset: two VGPRs and the
    // 64-bit address arithmetic saved, 0.5 % faster) were tried in round 2 and are WRONG for this kernel: the counted vmcnt
    // waits rely on vector-memory operations retiring in issue order, which holds among GLOBAL operations (the LDS-DMA loads
    // and these stores) but not between MUBUF and GLOBAL ones -- with buffer stores the counter reached its target while a
    // DMA piece was still in flight and the next step read a stale ring row (caught by the libhadi_strict.so comparison
    // and the oracle tests at 2 and 8 nodes per lane).
    hadi_put_block<B, G, T>(c.Yi + (size_t)j * rowp, half, lane, yo);
    if (lane == 0 && first_half) c.Yi[(size_t)j * rowp + c0slot] = (T)yout_c0;
    HADI_STAMPC(29);  // final correction + store issue
}

// LDS: [HADI_STRIP_WAVES wavefronts][4 ring slots][rowp] + the 4 s-coefficient arrays.  Grid = n_inst * sblocks blocks.
// T = float: fp32-state sweep (European only), as in hadi_pass_a.
// G = 2 (512 < m1 <= 1024, European): the 8 wavefronts form 4 PAIRS, each pair walks one strip, wavefront h of the pair owns
// half h of every row (its own pieces of the pair's ring slot, fetched by its own LDS-DMA and retired by its own counted
// wait -- no wavefront ever reads ring data its partner fetched, except the one boundary node, see below).  LDS:
// [4 pairs][NS slots][rowp] + 4 coefficient arrays of 1024 + the pairs' exchange buffers; with an fp64 state only NS = 3
// slots fit the 160 KB (rows j+1, j+2 landed, j+3 in flight), with an fp32 state 4 as above.
// MODE 1 / 2: predictor / corrector row pass of a Craig-Sneyd step (European, fp64 state).  The corrector's R1 and C2 rows are
// ordinary register loads (hadi_cs_row_load): the row of step t + 1 is requested at the top of step t and used a step later.
template <int B, int AMER, class T = double, int G = 1, int MODE = 0>
#ifndef HADI_STRIP_OCC_B4
#define HADI_STRIP_OCC_B4 2
#endif
// (2 nodes per lane, American P representation: at 4 waves per SIMD -- 128 VGPRs -- the kernel spills two registers, and a
// scratch reload inside the row loop drains the DMA prefetch: 3 there)
__global__ void __launch_bounds__(64 * HADI_STRIP_WAVES(B), (B >= 8 ? 2 : B == 4 ? HADI_STRIP_OCC_B4 : AMER == 2 ? 3 : 4)) hadi_pass_a_strip(HadiSweepArgs a, int n) {
    static_assert(sizeof(T) == 8 || AMER == 0, "the fp32-state sweep is European only");
    static_assert(G == 1 || (G == 2 && B == 8), "paired strips: 8 nodes per lane");
    static_assert(MODE == 0 || (AMER == 0 && sizeof(T) == 8), "Craig-Sneyd: European sweeps, fp64 state");
    static_assert(MODE != 3 || G == 2, "MODE 3: the coupling-column table of the paired strips");
    // paired strips: the pair's coupling column from the table MODE 3 built at the start of the solve (hadi_strip_step, RSTAB)
#ifndef HADI_PAIR_RSTAB
#define HADI_PAIR_RSTAB 1
#endif
    // (not the P representation: that kernel sits at 256 VGPRs, and the two registers the table entry keeps across the step
    // sent six others to scratch -- a scratch reload inside the row loop drains the DMA prefetch)
    // ... and not the explicit (U, lambda_bar) pair either: measured 3 % SLOWER there (its lambda_bar rows are register loads of the
    // compiler already; one more shifts its waits) -- European sweeps, both state precisions (profiles/r04_pair_spike_ab.txt)
    constexpr bool RSTAB = (G == 2 && MODE == 0 && AMER == 0 && HADI_PAIR_RSTAB);
    HADI_DYN_SMEM(double, smem);
    constexpr int NS = HADI_STRIP_NS(B, G, (int)sizeof(T)), NWV = HADI_STRIP_WAVES(B), NPAIR = NWV / G, c0slot = 64 * B * G;
    // American P representation at 8 nodes per lane: one slot stays BEHIND the prefetch -- row j itself, whose raw P the step
    // reads again for lambda_bar -- so the row D = NS - 1 ahead is fetched, not the row NS ahead
    constexpr int KEEP = (AMER == 2 && B >= HADI_AMP_KEEP_MIN_B && G == 1) ? 1 : 0, D = NS - KEEP;
    constexpr int NA = D - 2;  // DMA batches in flight behind the one that is waited for
    const int lane = threadIdx.x & 63;
    const int wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    const int pair = wave / G, half = wave - pair * G;  // (G = 1: pair = wave, half = 0)
    const int total = a.n_inst * a.sblocks;
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);
    if (logical >= total) return;
    const int inst = logical / a.sblocks, sb = logical - inst * a.sblocks;
    // The instance's parameter block is REQUESTED here and CONSUMED behind the prologue's row fetches (round 4): consumed at
    // once -- `if (n > ip.N) return` -- its memory round trip (1 - 2 us, one per launch and wavefront, nothing to overlap it
    // with) stood in front of every other load of the prologue; a launch of short strips is mostly prologue.
    const HadiInstPar ip = a.ipar[inst];
    const int nrows = a.L.nrows, npad = a.L.nrows_pad, rowp = a.L.rowp;
    double *coef = reinterpret_cast<double *>(reinterpret_cast<T *>(smem) + (size_t)NPAIR * NS * rowp);
    // P representation: the payoff row (it depends on s only: v-row 0 of the packed payoff) behind the coefficient arrays;
    // re-read from LDS every row rather than held in 2 B registers per lane (that version spilled)
    const double *payl = coef + 4 * 64 * B * G;
    const int j0 = (sb * NPAIR + pair) * a.RS;
    const bool has_strip = j0 < nrows;  // (wave-uniform; a wavefront without a strip only helps with the shared copies below)
    const int j1 = (j0 + a.RS < nrows) ? j0 + a.RS : nrows;

    HadiStripCtxT<T> c;
    c.lane = lane;
    c.rowp = rowp;
    c.coef = coef;
    c.half = half;
    double *const xch0 = coef + 4 * 64 * B * G + (AMER == 2 ? rowp : 0);  // the pairs' exchange buffers (behind the payoff row)
    c.xch = xch0 + pair * 16;
    c.err = a.err; c.debug = a.debug;
    const T *__restrict__ Ub = reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride;
    c.Yi = reinterpret_cast<T *>(a.Y) + (size_t)inst * a.L.inst_stride;
    c.Li = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride : nullptr;
    c.R1i = MODE ? a.R1 + (size_t)inst * a.L.inst_stride : nullptr;
    c.C2i = MODE ? a.C2 + (size_t)inst * a.L.inst_stride : nullptr;
    c.b2r = a.b2row + (size_t)inst * rowp;
    // P representation: 1/dt, and which node is s_max (lambda_bar stays 0 there, as in hadi_row_step)
    c.inv_dt = 0.0; c.m1_lane = -1; c.m1_r = -1;
    if constexpr (AMER == 2) {
        const int e1 = a.L.m1 - 1;  // node i = m1 is element m1 - 1 of the row's 64 B G interior nodes
        if (e1 / (64 * B) == half) {
            c.m1_lane = (e1 - half * 64 * B) / B;
            c.m1_r = (e1 - half * 64 * B) % B;
        }
    }

    T *ring = reinterpret_cast<T *>(smem) + (size_t)pair * NS * rowp;
    auto slot = [&](int jj) { return ring + (size_t)((NS & (NS - 1)) == 0 ? (jj & (NS - 1)) : (jj + 12) % NS) * rowp; };  // (jj >= -4)
    // returns the number of vector-memory instructions issued (rows outside the allocation are zero-filled)
    auto fetch = [&](int jj) -> int {
        const bool exists = jj >= 0 && jj < npad;
        if constexpr (G > 1) {
            return hadi_half_row_to_lds<B, T>(Ub + (ptrdiff_t)jj * rowp, slot(jj), half, lane, exists);
        } else {
            hadi_row_to_lds_fixed<B, T>(Ub + (ptrdiff_t)jj * rowp, slot(jj), lane, exists);
            return exists ? hadi_row_dma_count<T>(rowp) : 0;
        }
    };
    // Direction of the walk: even strips go up (j0 -> j1-1), odd strips come down (j1-1 -> j0).  Neighbouring strips
    // then touch their shared halo rows at the same time -- both start there or both end there -- so the second reader
    // finds them in L2 instead of fetching them again ~100 us later (HBM reads of this pass 9.8 -> ~9 B per node).
    // Below, "behind" = rows already passed (registers), "ahead" = rows still to come (LDS ring / in flight); for a
    // descending strip the row-table scalars of the +1/+2 and -1/-2 neighbours simply swap roles.
    // Round 4: EVEN strips come down, ODD strips go up, so the strips 2k and 2k+1 -- always in the same block -- START on
    // either side of their common boundary.  What one of them has behind it at the start (two rows) is what the other starts
    // on and has one ahead: those rows are read from the partner's ring after the prologue's barrier instead of from memory a
    // second time, and the strip's own first row comes through its ring as well (`shared` below): cnt + 2 rows per strip by
    // LDS-DMA and nothing else, where cnt + 1 + 3 were read.
    const int sidx = sb * NPAIR + pair;
    const int dir = (sidx & 1) ? 1 : -1;
    const int cnt = j1 - j0;
    const int js = dir > 0 ? j0 : j1 - 1;
    auto row_ok = [&](int jj) { return jj >= 0 && jj < npad; };
    // (wave-uniform) the partner strip exists; a last strip without one keeps the register loads of its rows behind
    const bool shared = has_strip && (sidx ^ 1) * a.RS < nrows;
    const T *pring = reinterpret_cast<T *>(smem) + (size_t)(pair ^ 1) * NS * rowp;
    auto pslot = [&](int jj) { return pring + (size_t)((NS & (NS - 1)) == 0 ? (jj & (NS - 1)) : (jj + 12) % NS) * rowp; };
    // ---- prologue: the next rows ahead to the ring, the two rows behind and the first row to registers ----
    // aft[k] = vector-memory instructions issued after the DMA of the row 2 + k ahead: aft[0] belongs to the row that is
    // waited for next, the row NS - 1 ahead is the youngest DMA (nothing behind it yet)
    int aft[NA];
#pragma unroll
    for (int k = 0; k < NA; k++) aft[k] = 0;
    // Order of the prologue: this wavefront's row fetches (LDS-DMA) and register loads are ISSUED first, then the block
    // copies the shared s-coefficient arrays (global -> LDS) and meets at the only block-wide barrier -- the two memory
    // round trips overlap instead of following each other (a launch of short strips is mostly prologue: 64 instances of
    // 512x256, 9-row strips: 0.0380 -> see DESIGN.md section 5).
    if (has_strip) {
        if (KEEP || shared) fetch(js);  // (the first row too: the step reads its raw P from the ring / `shared` above)
        fetch(js + dir);
        fetch(js + 2 * dir);
#pragma unroll
        for (int q = 3; q < D; q++) {
            const int zq = fetch(js + q * dir);
#pragma unroll
            for (int k = 0; k < NA; k++)
                if (k + 2 < q) aft[k] += zq;
        }
    }
    double rs_next = 0.0;  // RSTAB: the table entry of the NEXT step's row (an ordinary load, requested a step ahead: hadi_cs_row_load)
    const double *rs_lane = (RSTAB || MODE == 3) ? a.rs_tab + ((size_t)inst * nrows * 2 + half) * 64 + lane : nullptr;  // + 128 j
    if constexpr (RSTAB) rs_next = hadi_nt_load(rs_lane + (size_t)(has_strip ? js : 0) * 128);
    HadiCsRow<B> cs_next;  // MODE 2: R1 / C2 of the NEXT step's row (requested a step ahead)
    if constexpr (MODE == 2) {
        const size_t ro = (size_t)(has_strip ? js : 0) * rowp;
        hadi_cs_row_load<B, G>(c.R1i + ro, c.C2i + ro, half, lane, cs_next);
    }
    // rows behind by 2, behind by 1 (carried in the state's own type: with an fp32 state they are exact floats and cost
    // half the registers), current row (double: used throughout the step)
    T um2[B], um1[B];
    double u0[B];
    // The i = 0 column of the five stencil rows is wave-uniform: ONE register pair carries it, spread over the lanes
    // (lane k = row j - 2 + k in walking order), read with v_readlane where needed and shifted by a DPP move per step.
    // G = 2: the low half owns the i = 0 column; `evec` carries, the same way, the partner's node next to this half (the
    // high half's first node for the low half and vice versa) on the rows behind / at / ahead of j (lanes 1, 2, 3).
    double c0vec, evec = 0.0;
    int epos = 0;
    if constexpr (G > 1) {
        const int inode = (half == 0) ? 64 * B + 1 : 64 * B;
        epos = (sizeof(T) == 4) ? hadi_pos_f32(B, G, inode) : hadi_pos(B, G, inode);
    }
    double t2[B], t1[B];
#pragma unroll
    for (int r = 0; r < B; r++) t2[r] = t1[r] = u0[r] = 0.0;
    c0vec = 0.0;
    if (has_strip) {
        if (!shared) {
            if (row_ok(js - 2 * dir)) hadi_get_block<B, G, T>(Ub + (ptrdiff_t)(js - 2 * dir) * rowp, half, lane, t2);
            if (row_ok(js - dir)) hadi_get_block<B, G, T>(Ub + (ptrdiff_t)(js - dir) * rowp, half, lane, t1);
            hadi_get_block<B, G, T>(Ub + (size_t)js * rowp, half, lane, u0);
        }
        const int rr = js + (lane - 2) * dir;
        c0vec = (half == 0 && lane < 4 && row_ok(rr)) ? (double)Ub[(ptrdiff_t)rr * rowp + c0slot] : 0.0;
        if constexpr (G > 1) evec = (lane < 4 && row_ok(rr)) ? (double)Ub[(ptrdiff_t)rr * rowp + epos] : 0.0;
    }
    // the shared arrays' global loads are issued before the parameter block is consumed as well: the scaling by it and the
    // LDS stores follow below (two dependent memory round trips of the prologue become one)
    constexpr int NCOPY = (4 * 64 * B * G) / (64 * NWV);
    static_assert(NCOPY * 64 * NWV == 4 * 64 * B * G, "coefficient arrays: whole rounds of the block");
    double sc_tmp[NCOPY];
    {
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 64 * B * G;
#pragma unroll
        for (int q = 0; q < NCOPY; q++) sc_tmp[q] = sc[threadIdx.x + q * 64 * NWV];
    }
    // ---- the instance's parameters are consumed here, behind the row fetches (see the top) ----
    if (n > ip.N) {  // (block-uniform: this instance has fewer time steps -- multi-maturity batches)
#if !defined(HADI_EMU)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of this wavefront may outlive it
#endif
        return;
    }
    c.dt = hadi_uniform_d(ip.dt); c.thdt = hadi_uniform_d(ip.thdt);
    c.c1 = hadi_uniform_d(1.0 + ip.thdt * ip.half_rd);
    c.kap = hadi_uniform_d((ip.dt - ip.thdt) / ip.thdt);  // the host keeps theta = 0 off this kernel
    c.hr0 = hadi_uniform_d(ip.hr0); c.inv0 = hadi_uniform_d(1.0 / (1.0 + ip.thdt * ip.hr0));
    c.e_nm1 = hadi_uniform_d(exp(ip.bc_rate * ip.dt * (n - 1)));  // device_solver.hpp:238
    c.e_n = hadi_uniform_d(exp(ip.bc_rate * ip.dt * n));          // device_solver.hpp:246
    if constexpr (AMER == 2) c.inv_dt = hadi_uniform_d(1.0 / ip.dt);
    {   // s-coefficient arrays to LDS; the two beta arrays scaled by -theta dt (r_d - r_f) on the way (hadi_strip_step)
        const double mq = -(ip.thdt * ip.q);
#pragma unroll
        for (int q = 0; q < NCOPY; q++) {
            const int e = threadIdx.x + q * 64 * NWV;
            coef[e] = (e < 2 * 64 * B * G) ? mq * sc_tmp[q] : sc_tmp[q];
        }
    }
    if constexpr (AMER == 2) {
        const double *__restrict__ pg = a.U0 + (size_t)inst * a.L.inst_stride;
        double *pw = coef + 4 * 64 * B * G;
        for (int e = threadIdx.x; e < rowp; e += 64 * NWV) pw[e] = pg[e];
    }
    if constexpr (G > 1) {  // the pairs' exchange buffers (values + rendezvous tokens, all zero: no row has token 0)
        if (threadIdx.x < NPAIR * 16) xch0[threadIdx.x] = 0.0;
    }
    hadi_wait_vmcnt(0);  // this wavefront's prologue rows have landed (the partner reads two of them behind the barrier)
    __syncthreads();     // the coefficient arrays are shared
    if (shared) {
        hadi_get_block<B, G, T>(pslot(js - 2 * dir), half, lane, t2);  // = the partner's row one ahead
        hadi_get_block<B, G, T>(pslot(js - dir), half, lane, t1);      // = the partner's first row
        hadi_get_block<B, G, T>(slot(js), half, lane, u0);
#if !defined(HADI_EMU)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    }
    __syncthreads();  // ... and nobody's first fetch of the loop lands in a slot its partner is still reading
    if (!has_strip) return;
    // 8 nodes per lane, European fp64 (the headline kernel): two of the four arrays fit the registers left over (224 -> 250
    // VGPRs, no spill): 8 of the 16 coefficient reads per row step less on the LDS pipe, +0.7 % on 512x256 x256 (three
    // interleaved runs of each build on one box, gpurun_out/r03aa); 3: only the last array (no gain measured)
#ifndef HADI_STRIP_CREG8
#define HADI_STRIP_CREG8 2
#endif
    constexpr int CREG = (B <= HADI_STRIP_CREG_MAX_B && G == 1) ? 1 : (B == 8 && G == 1 && AMER == 0 && sizeof(T) == 8 && MODE == 0) ? HADI_STRIP_CREG8 : 0;
    double cf[4 * B];
    if constexpr (CREG) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            double t[B];
            hadi_get_block<B, 1>(coef + q * 64 * B, 0, lane, t);
#pragma unroll
            for (int r = 0; r < B; r++) cf[q * B + r] = t[r];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4 * B; e++) cf[e] = 0.0;
    }
    if constexpr (AMER == 2) {  // U = max(P, U_0) on the rows behind (the current row keeps its raw P for lambda_bar)
        double pay[B];
        hadi_get_block<B, G>(payl, half, lane, pay);
#pragma unroll
        for (int r = 0; r < B; r++) {
            t2[r] = fmax(t2[r], pay[r]);
            t1[r] = fmax(t1[r], pay[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < B; r++) {
        um2[r] = (T)t2[r];
        um1[r] = (T)t1[r];
    }
#if !defined(HADI_EMU)
    // Consume the prologue's register loads HERE: otherwise hipcc parks their s_waitcnt vmcnt(0) at the loop header,
    // where it would retire the DMA prefetch and the result stores in every iteration.
#pragma unroll
    for (int r = 0; r < B; r++) asm volatile("" : "+v"(um2[r]), "+v"(um1[r]), "+v"(u0[r]));  // (T and double operands)
    asm volatile("" : "+v"(c0vec));
    if constexpr (G > 1) asm volatile("" : "+v"(evec));
#endif

#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[32] = {0};
    c.stamp_acc_ = stamp_store_;
#endif
    HADI_STAMP_DECL(c.stamp_acc_)
    for (int t = 0; t < cnt; t++) {
        const int j = js + dir * t;
        HADI_STAMPC(30);  // carry + loop
        HadiSRow srow;
        // (round 4, measured and dropped: requesting the entry one step AHEAD -- at the end of the step before, the first one in
        // the prologue -- keeps 24 scalar registers live across the loop edge; at 106 SGPRs the compiler parks them in VGPR
        // lanes (two variants even spill to scratch): row pass +1.2 % at 33-row strips, +3 % at 9 rows, +6.5 % on paired strips)
        hadi_sload_issue(a.rowc + ((size_t)inst * nrows + j) * HADI_RC + HADI_SRC0, srow);  // flies during the DMA wait
        double rs_cur = 0.0;
        if constexpr (RSTAB) {
            rs_cur = rs_next;  // (the compiler's wait for the load of a step ago sits here, in front of this step's loads and DMA)
            int nl = 0;
#if !defined(HADI_EMU)
            asm volatile("" : "+v"(rs_cur) :: "memory");
#endif
            if (t + 1 < cnt) {
                rs_next = hadi_nt_load(rs_lane + (size_t)(j + dir) * 128);  // (non-temporal: 33 MB of table per step must not displace Y from the memory-side cache -- the column pass ran 4 % slower with default-policy loads here)
                nl = 1;
            }
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");
#endif
#pragma unroll
            for (int k = 0; k < NA; k++) aft[k] += nl;
        }
        HadiCsRow<B> csrow;
        if constexpr (MODE == 2) {
            csrow = cs_next;  // (the compiler's own wait for the loads of a step ago sits in front of their first use)
            int nl = 0;
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");  // the next row's loads stay HERE: behind the stores of the step before, ahead of this step's DMA
#endif
            if (t + 1 < cnt) {
                const size_t ro = (size_t)(j + dir) * rowp;
                nl = hadi_cs_row_load<B, G>(c.R1i + ro, c.C2i + ro, half, lane, cs_next);
            }
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");
#endif
#pragma unroll
            for (int k = 0; k < NA; k++) aft[k] += nl;  // (they come behind every DMA batch in flight)
        }
        hadi_wave_rendezvous();
        // the row D ahead goes to the slot of row j (of row j - 1 when one slot is kept behind): that row is in registers,
        // and this wavefront's last read of the slot (in the previous step) has been retired there.  Issued BEFORE the
        // wait below, so that the prefetch does not queue behind it.
        int z = 0;
        if (t + D <= cnt + 1) z = fetch(j + D * dir);
        hadi_wait_vmcnt(aft[0] + z);  // the row two ahead has landed (the row one ahead landed a step earlier)
        HADI_STAMPC(24);  // wait for the DMA
#pragma unroll
        for (int k = 0; k + 1 < NA; k++) aft[k] = aft[k + 1] + z;
        aft[NA - 1] = 0;
        hadi_wave_rendezvous();
        double up1[B], up2[B];
        hadi_get_block<B, G, T>(slot(j + dir), half, lane, up1);
        hadi_get_block<B, G, T>(slot(j + 2 * dir), half, lane, up2);
        if (half == 0) {  // (wave-uniform; always true for G = 1)
            const double c0new = (double)slot(j + 2 * dir)[c0slot];  // (every lane reads the same word)
            c0vec = (lane == 4) ? c0new : c0vec;
        }
        double rt[HADI_RCL];
        hadi_sload_wait(srow, rt);  // one lgkmcnt(0) for the table entry and the LDS reads above
        if (dir < 0) {  // descending: "behind" rows are j+1, j+2 -- swap the neighbour weights instead of the arrays
            double w;
            w = rt[RC_WMS - HADI_SRC0]; rt[RC_WMS - HADI_SRC0] = rt[RC_WPS - HADI_SRC0]; rt[RC_WPS - HADI_SRC0] = w;
            w = rt[RC_L2 - HADI_SRC0]; rt[RC_L2 - HADI_SRC0] = rt[RC_U2 - HADI_SRC0]; rt[RC_U2 - HADI_SRC0] = w;
            w = rt[RC_L1 - HADI_SRC0]; rt[RC_L1 - HADI_SRC0] = rt[RC_U1 - HADI_SRC0]; rt[RC_U1 - HADI_SRC0] = w;
        }
        HADI_STAMPC(25);  // LDS reads + table entry + DMA issue
        double praw[B], lamc0 = 0.0;
        const double c0m2 = hadi_read_lane(c0vec, 0), c0m1 = hadi_read_lane(c0vec, 1), c00 = hadi_read_lane(c0vec, 2);
        const double c0p1 = hadi_read_lane(c0vec, 3), c0p2 = hadi_read_lane(c0vec, 4);
        double e0m2 = c0m2, e0m1 = c0m1, e00 = c00, e0p1 = c0p1, e0p2 = c0p2;  // (the carried i = 0 values stay raw)
#pragma unroll
        for (int r = 0; r < B; r++) praw[r] = 0.0;
        if constexpr (AMER == 2) {
            double pay[B];
            hadi_get_block<B, G>(payl, half, lane, pay);
            const double pay_c0 = payl[c0slot];
#pragma unroll
            for (int r = 0; r < B; r++) {
                if constexpr (G == 2) {
                    // paired strips: u0 stays the raw P (hadi_strip_step, RAW_U0); the row behind was carried raw as well
                    um1[r] = (T)fmax((double)um1[r], pay[r]);
                } else {
                    if constexpr (!KEEP) praw[r] = u0[r];  // the raw P of row j: lambda_bar comes from it inside the step
                    u0[r] = fmax(u0[r], pay[r]);
                }
                up1[r] = fmax(up1[r], pay[r]);
                up2[r] = fmax(up2[r], pay[r]);
            }
            lamc0 = fmax(0.0, (pay_c0 - c00) * c.inv_dt);
            e0m2 = fmax(c0m2, pay_c0); e0m1 = fmax(c0m1, pay_c0); e00 = fmax(c00, pay_c0);
            e0p1 = fmax(c0p1, pay_c0); e0p2 = fmax(c0p2, pay_c0);
        }
        double dm2[B], dm1[B];
#pragma unroll
        for (int r = 0; r < B; r++) {
            dm2[r] = (double)um2[r];
            dm1[r] = (double)um1[r];
        }
        double un[B];
        double xb_ = 0.0, x0_ = 0.0, xa_ = 0.0;  // the partner's boundary node on the rows behind / at / ahead (G = 2)
        if constexpr (G > 1) {
            xb_ = hadi_read_lane(evec, 1); x0_ = hadi_read_lane(evec, 2); xa_ = hadi_read_lane(evec, 3);
            if constexpr (AMER == 2) {  // (the carried values stay raw P: U = max(P, U_0) on the partner's node too)
                const double pay_e = payl[epos];
                xb_ = fmax(xb_, pay_e); x0_ = fmax(x0_, pay_e); xa_ = fmax(xa_, pay_e);
            }
        }
        double *rs_out = (MODE == 3) ? const_cast<double *>(rs_lane) + (size_t)j * 128 : nullptr;
        if (MODE < 2 && j == nrows - 1) hadi_strip_step<B, AMER, (MODE < 2), T, G, CREG, MODE, RSTAB>(c, j, rt, dm2, dm1, u0, up1, up2, e0m2, e0m1, e00, e0p1, e0p2, praw, lamc0, slot(j + dir), un, xb_, x0_, xa_, slot(j), payl, cf, &csrow, rs_cur, rs_out);
        else hadi_strip_step<B, AMER, false, T, G, CREG, MODE, RSTAB>(c, j, rt, dm2, dm1, u0, up1, up2, e0m2, e0m1, e00, e0p1, e0p2, praw, lamc0, slot(j + dir), un, xb_, x0_, xa_, slot(j), payl, cf, &csrow, rs_cur, rs_out);
        // the row's vector stores (the i = 0 stores are not counted: lower bound); the predictor stores R1 and C2 as well; the
        // table build (MODE 3) stores one double per lane (counted as nothing: lower bound)
        constexpr int NST = (MODE == 3 ? 0 : MODE == 1 ? 3 : 1) * hadi_put_block_stores<B, T>();
#pragma unroll
        for (int k = 0; k < NA; k++) aft[k] += NST;
        double enew = 0.0;
        if constexpr (G > 1) {
            // The partner's boundary node of the row TWO ahead, from the partner's half of the ring slot.  Safe here and only
            // here: the partner retired its DMA of that row before it published this step's token (which the exchange inside
            // the step has just seen), and it refills that slot two steps on -- after the next exchange, which needs this
            // wavefront's next token.
            enew = (double)slot(j + 2 * dir)[epos];  // (every lane reads the same word)
        }
#pragma unroll
        for (int r = 0; r < B; r++) {
            um2[r] = um1[r];
            um1[r] = (T)u0[r];
            u0[r] = un[r];
        }
#if !defined(HADI_EMU)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the re-read is retired before the next step reuses that slot
#endif
        c0vec = hadi_lane_next(c0vec);  // lane k takes lane k + 1: one row on
        if constexpr (G > 1) {
            evec = hadi_lane_next(evec);
            evec = (lane == 3) ? enew : evec;
        }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev_) :: "memory");  // the step stamped itself
#endif
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (HADI_STAMPS == 4 && lane == 0)
        for (int k = 24; k < 31; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
}

// ------------------------------------------------------------------------------------------------
// Pair strips (128 < m1 <= 256): TWO strips per wavefront, 32 lanes x 8 nodes each.
// At 4 nodes per lane the strip kernel spends as many instructions on a 256-node row as the 8-node kernel on a 512-node one
// in everything that is per LANE rather than per node -- the six levels of cyclic reduction, the row scalars, the ring
// bookkeeping: 380 VALU + 133 SALU + 70 LDS instructions per row and wavefront (PMC, profiles/r02_c3_pmc_summary.json)
// against 513 + ~150 + ~90 for twice the nodes, and the SIMDs are issue-bound (two wavefronts, 108 % of the issue cycles
// active).  Here a wavefront runs the 8-node arithmetic on TWO independent strips at once: lanes 0..31 walk strip A, lanes
// 32..63 strip B of the same instance, lane h of a half owning the nodes 8h+1 .. 8h+8 of its strip's current row.  Same
// storage layout as the 4-node kernels (row of 272 doubles: the column pass does not change), so a lane's nodes are two
// 32-byte chunks of a row: {0,1,4,5} and {2,3,6,7}.
//   cross-lane traffic   the distance-1 exchanges are the same wave shifts; what crosses the half boundary (lane 31 <-> 32)
//                        is multiplied by a zero coefficient on either side (node 256 or a pad has no upper neighbour: Bp =
//                        Dp = 0; the first node's coupling to i = 0 is moved to the right-hand side), so no fix-up is needed;
//                        the cyclic reduction has five levels, its permutes stay inside the half
//   row scalars          differ between the halves: both halves' table entries come through the scalar cache as before (two
//                        sets of SGPRs, issued at the loop top) and are moved to per-lane registers under the halves' exec
//                        masks -- 40 moves per step; vector loads of the entries (tried first) kept 48 more registers live
//                        across the step and the kernel spilled
//   ring                 a slot holds the two rows of a step interleaved in 512-byte pieces [piece][half] (one LDS-DMA
//                        instruction moves a piece of BOTH rows: lanes 0..31 from row A, 32..63 from row B), the two 128-byte
//                        tails adjacent: 4352 B per slot, 6 DMA instructions per step
//   out-of-range rows    the strips of a wavefront are equally long or the second is shorter / empty; a finished or empty
//                        half keeps computing on clamped rows and stores nothing.  Rows j-2 .. j+2 outside the grid are
//                        clamped too: they only ever meet zero weights (as in hadi_small_seq_kernel)
// Counted waits as in hadi_pass_a_strip: the row two ahead has landed, younger DMA batches and the result stores stay in flight.
#define HADI_PAIR_SLOT 544   // doubles per ring slot
#define HADI_PAIR_WAVES 4    // wavefronts (= 8 strips) per block

HADI_DEV HADI_FORCEINLINE void hadi_pair_get(const double *p, int ch1, double (&u)[8]) {
    const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 2);
    const double2 c = *reinterpret_cast<const double2 *>(p + ch1), d = *reinterpret_cast<const double2 *>(p + ch1 + 2);
    u[0] = a.x; u[1] = a.y; u[4] = b.x; u[5] = b.y; u[2] = c.x; u[3] = c.y; u[6] = d.x; u[7] = d.y;
}
// The same from LDS, where the two 16-byte pieces of a chunk are SWAPPED for the lanes 8 .. 15 of every 16 (round 4).  A lane's
// chunk is 32 bytes, so 16 lanes reading "the first 16 bytes of my chunk" touch every other 16-byte bank group of two 256-byte
// bank rows: a 2-way conflict on every ds_read_b128 of this kernel (SQ_LDS_BANK_CONFLICT 5.0 M cycles per dispatch on config
// 3, as many as the reads themselves).  With the pieces of the upper eight lanes swapped in the LDS image, "first piece" reads
// hit the even groups in lanes 0 .. 7 and the odd groups in lanes 8 .. 15 -- all 16 distinct.  The image is swizzled where it
// is WRITTEN (the LDS-DMA's lane -> LDS position is fixed, its source address is not; the coefficient / payoff copies permute
// their index); the readers use two per-lane offsets instead of one offset and a constant.  offA / offB: where this lane's
// first / second piece sits (hadi_pair_offs).
HADI_DEV HADI_FORCEINLINE int hadi_pair_swz(int e) { return e ^ (((e >> 5) & 1) << 1); }  // position of element e (doubles) in a swizzled image
HADI_DEV HADI_FORCEINLINE void hadi_pair_offs(int h, int chunk, int &offA, int &offB) {
    const int s2 = (h & 8) ? 2 : 0;
    offA = chunk + s2;
    offB = chunk + 2 - s2;
}
HADI_DEV HADI_FORCEINLINE void hadi_pair_get_lds(const double *base, int offA, int offB, int ch1, double (&u)[8]) {
    const double2 a = *reinterpret_cast<const double2 *>(base + offA), b = *reinterpret_cast<const double2 *>(base + offB);
    const double2 c = *reinterpret_cast<const double2 *>(base + offA + ch1), d = *reinterpret_cast<const double2 *>(base + offB + ch1);
    u[0] = a.x; u[1] = a.y; u[4] = b.x; u[5] = b.y; u[2] = c.x; u[3] = c.y; u[6] = d.x; u[7] = d.y;
}
HADI_DEV HADI_FORCEINLINE void hadi_pair_put(double *p, const double (&u)[8]) {  // global row: the chunks are 128 doubles apart
    double2 a, b, c, d;
    a.x = u[0]; a.y = u[1]; b.x = u[4]; b.y = u[5]; c.x = u[2]; c.y = u[3]; d.x = u[6]; d.y = u[7];
    *reinterpret_cast<double2 *>(p) = a; *reinterpret_cast<double2 *>(p + 2) = b;
    *reinterpret_cast<double2 *>(p + 128) = c; *reinterpret_cast<double2 *>(p + 130) = d;
}
// LDS-DMA of the two rows `grow` (per lane: the row of this lane's half) into ring slot `slot`.  6 vector-memory instructions.
HADI_DEV HADI_FORCEINLINE void hadi_pair_fetch(const double *__restrict__ grow, double *slot, int lane) {
    const int H = lane >> 5, h = lane & 31;
#if defined(HADI_EMU)
    const int hs = h ^ (h >= 16 ? 1 : 0);  // (the piece this lane fetches: hadi_pair_get_lds)
    for (int pc = 0; pc < 4; pc++)
        for (int e = 0; e < 2; e++) slot[pc * 128 + H * 64 + 2 * h + e] = grow[64 * pc + 2 * hs + e];
    if (h < 8)
        for (int e = 0; e < 2; e++) slot[512 + 16 * H + 2 * h + e] = grow[256 + 2 * h + e];
#else
    const double *gsrc = grow + 2 * h;                        // (the tails: not swizzled)
    const double *gswz = grow + 2 * (h ^ (h >= 16 ? 1 : 0));  // the piece this lane fetches: hadi_pair_get_lds
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)slot);
#pragma unroll
    for (int pc = 0; pc < 4; pc++) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gswz + 64 * pc), "s"(lds0 + 1024u * pc) : "memory");
    }
    // the 128-byte tails (slot 256 = i = 0 and the pads): half A to bytes 4096.., half B right behind it (the hardware adds
    // 16 x lane to M0: 512 for lane 32, hence the base 4096 + 128 - 512)
    if (lane < 8) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc + 256), "s"(lds0 + 4096u) : "memory");
    }
    if (lane >= 32 && lane < 40) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc + 256), "s"(lds0 + 3712u) : "memory");
    }
#endif
}
#define HADI_PAIR_DMA 6

// One step of both strips of the wavefront.  rv: this lane's row scalars (entries RC_L2 .. RC_WPS of its half's row); the
// stencil rows as in hadi_strip_step (B = 8); c0*: the i = 0 column of the five rows of this lane's half; yrow: this half's
// output row (global); act: this half still has rows (stores are skipped otherwise).
template <int AMER, bool LAST>
HADI_DEV HADI_FORCEINLINE void hadi_pair_step(const HadiStripCtxT<double> &c, int h, bool act, bool is_last, const double (&rv)[HADI_RCL],
                                              const double (&um2)[8], const double (&um1)[8], const double (&u0)[8],
                                              const double (&up1)[8], const double (&up2)[8], double c0m2, double c0m1, double c00,
                                              double c0p1, double c0p2, double lamc0_in, const double *raw_slot,
                                              const double *next_slot, int offA, int offB, double (&u_next)[8], double *yrow,
                                              const double *lrow) {
    constexpr int B = 8, NB = 7, c0slot = 256;
    const int lane = c.lane;
    const bool first = (h == 0);
    const double dt = c.dt, thdt = c.thdt, c1 = c.c1, kap = c.kap, e_nm1 = c.e_nm1, e_n = c.e_n;
    const double vth = rv[RC_VTH - HADI_SRC0];
    const double wm = rv[RC_WMS - HADI_SRC0], wz = rv[RC_WZS - HADI_SRC0], wp = rv[RC_WPS - HADI_SRC0];
    const double a2l2 = rv[RC_L2 - HADI_SRC0], a2l1 = rv[RC_L1 - HADI_SRC0], a2m = rv[RC_M - HADI_SRC0], a2u1 = rv[RC_U1 - HADI_SRC0],
                 a2u2 = rv[RC_U2 - HADI_SRC0];
    const double b1val = rv[RC_B1VAL - HADI_SRC0];
    const int b1raw = (int)rv[RC_B1COL - HADI_SRC0];
    const bool b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;
    const int b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
    const int b1e = b1col - 1;
    const int b1k = (b1col >= 1 && (b1e >> 3) == h) ? (b1e & 7) : -1;  // the node of this lane that carries the row's b1 entry

    // ---- column i = 0 ----------------------------------------------------------------------------------
    const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c00 + a2u1 * c0p1 + a2u2 * c0p2;
    const double b1c0 = b1_at0 ? b1val : 0.0;
    const double b2c0 = (LAST && is_last) ? c.b2r[c0slot] : 0.0;
    const double lamc0 = (AMER == 1) ? lrow[c0slot] : (AMER == 2) ? lamc0_in : 0.0;
    const double a1c0 = -c.hr0 * c00;
    double y0c0 = c00 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1 + lamc0);
    y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
    const double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
    const double x0 = y0c0 * c.inv0;
    const double yout_c0 = x0 + c2c0;

    // ---- explicit operators ---------------------------------------------------------------------------------
    double tt[B], A2U[B];
#pragma unroll
    for (int r = 0; r < B; r++) {
        tt[r] = wm * um1[r] + wz * u0[r] + wp * up1[r];
        A2U[r] = a2l1 * um1[r] + a2m * u0[r] + a2u1 * up1[r];
    }
#pragma unroll
    for (int r = 0; r < B; r++) A2U[r] = fma(a2l2, um2[r], A2U[r]);
#pragma unroll
    for (int r = 0; r < B; r++) A2U[r] = fma(a2u2, up2[r], A2U[r]);
    double u0L = hadi_lane_prev(u0[B - 1]), tL = hadi_lane_prev(tt[B - 1]);
    const double u0R = hadi_lane_next(u0[0]), tR = hadi_lane_next(tt[0]);  // (lane 31: lane 32's values, times Bp = Dp = 0)
    if (first) {
        u0L = c00;
        tL = wm * c0m1 + wz * c00 + wp * c0p1;
    }
    const double b1add = b1val * (dt * e_nm1 + thdt * (e_n - e_nm1));

    double lam[B], b2v[B];
    if constexpr (AMER == 1) hadi_pair_get(lrow + 4 * h, 128, lam);
    if constexpr (AMER == 2) {
        double praw[B];
        hadi_pair_get_lds(raw_slot, offA, offB, 256, praw);  // the raw P of row j, still intact in its ring slot
#pragma unroll
        for (int r = 0; r < B; r++) {
            lam[r] = (u0[r] - praw[r]) * c.inv_dt;  // exactly max(0, (U_0 - P) / dt): U = max(P, U_0)
            if (h == c.m1_lane && r == c.m1_r) lam[r] = 0.0;
        }
    }
    if constexpr (LAST) {
#pragma unroll
        for (int r = 0; r < B; r++) b2v[r] = 0.0;
        if (is_last) hadi_pair_get(c.b2r + 4 * h, 128, b2v);
    }

    hadi_set_prio(1);
    double Bm[B], Bp[B], Dm[B], Dp[B];
    double ys[B], ps[B], gs[B], iu[B], cp[B];
    double il_last = 0.0, im_last = 1.0, d_last = 0.0;
    int cA, cB;
    hadi_pair_offs(h, 4 * h, cA, cB);
#pragma unroll
    for (int r = 0; r < B; r++) {
        if ((r & 3) == 0 || (r & 3) == 2) {
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");
#endif
            // nodes {0,1,4,5} sit in the lane's first chunk, {2,3,6,7} in the second (128 doubles on); the coefficient arrays
            // are swizzled images too (hadi_pair_get_lds)
            const int co = ((r & 4) ? cB : cA) + ((r & 2) ? 128 : 0);
            const double2 t0 = *reinterpret_cast<const double2 *>(c.coef + 0 * 256 + co);
            const double2 t1 = *reinterpret_cast<const double2 *>(c.coef + 1 * 256 + co);
            const double2 t2 = *reinterpret_cast<const double2 *>(c.coef + 2 * 256 + co);
            const double2 t3 = *reinterpret_cast<const double2 *>(c.coef + 3 * 256 + co);
            Bm[r] = t0.x; Bm[r + 1] = t0.y;
            Bp[r] = t1.x; Bp[r + 1] = t1.y;
            Dm[r] = t2.x; Dm[r + 1] = t2.y;
            Dp[r] = t3.x; Dp[r + 1] = t3.y;
        }
        const double uL = (r == 0) ? u0L : u0[r == 0 ? 0 : r - 1];
        const double uR = (r == B - 1) ? u0R : u0[r == B - 1 ? r : r + 1];
        const double tl = (r == 0) ? tL : tt[r == 0 ? 0 : r - 1];
        const double tr = (r == B - 1) ? tR : tt[r == B - 1 ? r : r + 1];
        double il = fma(-vth, Dm[r], Bm[r]);  // (Bm, Bp hold -theta dt (r_d - r_f) s beta_s: hadi_strip_step)
        iu[r] = fma(-vth, Dp[r], Bp[r]);
        const double sm = il + iu[r];
        const double im = c1 - sm;
        const double T1 = fma(-iu[r], uR, fma(-il, uL, fma(-im, u0[r], u0[r])));
        const double A0U = Bm[r] * tl - (Bm[r] + Bp[r]) * tt[r] + Bp[r] * tr;
        double S = A0U + A2U[r];
        if constexpr (LAST) S += b2v[r] * e_nm1;
        if constexpr (AMER) S += lam[r];
        double y = fma(dt, S, u0[r]);
        y = fma(kap, T1, y);
        y = fma(b1add, (b1k == r) ? 1.0 : 0.0, y);
        if (r == 0 && first) {  // x_0 is known: move it to the right-hand side
            y -= il * x0;
            il = 0.0;
        }
        if (r < NB) {
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
    double ra, rb, rcc, rf;
    {
        gs[NB - 1] = cp[NB - 1];
#pragma unroll
        for (int r = NB - 2; r >= 0; r--) {
            ys[r] = fma(-cp[r], ys[r + 1], ys[r]);
            ps[r] = fma(-cp[r], ps[r + 1], ps[r]);
            gs[r] = -cp[r] * gs[r + 1];
        }
        const double p0n = hadi_lane_next(ps[0]), g0n = hadi_lane_next(gs[0]), y0n = hadi_lane_next(ys[0]);
        ra = -il_last * ps[NB - 1];
        rb = im_last - il_last * gs[NB - 1] - iu[B - 1] * p0n;
        rcc = -iu[B - 1] * g0n;
        rf = d_last - il_last * ys[NB - 1] - iu[B - 1] * y0n;
    }
    {   // parallel cyclic reduction over the 32 interface unknowns of each half (normalised rows, see hadi_row_step)
        hadi_set_prio(3);
        const double rinv0 = hadi_rcp(rb);
        ra *= rinv0;
        rcc *= rinv0;
        rf *= rinv0;
#pragma unroll
        for (int s = 1; s < 32; s <<= 1) {
            const int up_lane = (lane & 32) | ((lane - s) & 31), dn_lane = (lane & 32) | ((lane + s) & 31);
            double aL, cL, fL, aR, cR, fR;
            if (s == 1) {
                aL = hadi_lane_prev(ra); cL = hadi_lane_prev(rcc); fL = hadi_lane_prev(rf);
                aR = hadi_lane_next(ra); cR = hadi_lane_next(rcc); fR = hadi_lane_next(rf);
            } else if (s == 16) {  // h - 16 and h + 16 are the same lane (mod 32)
                aL = aR = hadi_lane_get(ra, up_lane); cL = cR = hadi_lane_get(rcc, up_lane); fL = fR = hadi_lane_get(rf, up_lane);
            } else {
                aL = hadi_lane_get(ra, up_lane); cL = hadi_lane_get(rcc, up_lane); fL = hadi_lane_get(rf, up_lane);
                aR = hadi_lane_get(ra, dn_lane); cR = hadi_lane_get(rcc, dn_lane); fR = hadi_lane_get(rf, dn_lane);
            }
            const double bn = fma(-rcc, aR, fma(-ra, cL, 1.0));
            const double rn = hadi_rcp(bn);
            rf = fma(-rcc, fR, fma(-ra, fL, rf)) * rn;
            if (s < 16) {
                const double an = -(ra * aL) * rn;
                const double cn = -(rcc * cR) * rn;
                ra = an;
                rcc = cn;
            }
        }
    }
    hadi_set_prio(0);
    hadi_pair_get_lds(next_slot, offA, offB, 256, u_next);  // the row ahead again from its ring slot (flies during the stores)
    const double X = rf, XL = hadi_lane_prev(X);  // (lane 32: lane 31's X, times ps = 0)
    double yo[B];
#pragma unroll
    for (int r = 0; r < B; r++) {
        double x;
        if (r < NB) x = ys[r] - XL * ps[r] - X * gs[r];
        else x = X;
        double corr;
        if constexpr (LAST) corr = thdt * (b2v[r] * e_n - (A2U[r] + b2v[r] * e_nm1));
        else corr = -thdt * A2U[r];
        yo[r] = x + corr;
    }
    // (plain global stores: the counted waits need them in the same in-order queue as the LDS-DMA loads, hadi_strip_step)
    if (act) {
        hadi_pair_put(yrow + 4 * h, yo);
        if (first) yrow[c0slot] = yout_c0;
    }
}
#define HADI_PAIR_STORES 4  // vector stores per step counted by the waits (the i = 0 store is not: lower bound)

// LDS: [4 wavefronts][NS slots][544] | 4 coefficient arrays of 256 | payoff row of 272 (AMER == 2) | i = 0 history [4][2][4].
template <int AMER>
__global__ void __launch_bounds__(64 * HADI_PAIR_WAVES, 2) hadi_pass_a_pairs(HadiSweepArgs a, int n) {
    HADI_DYN_SMEM(double, smem);
    // 4-slot ring.  European / explicit pair: the rows 1 .. 4 ahead in the ring (1, 2 landed, 3, 4 in flight); P representation:
    // the slot of row j itself is kept for the step's re-read of the raw P, so the rows 1 .. 3 ahead.
    constexpr int NS = 4, D = (AMER == 2) ? 3 : 4, NWV = HADI_PAIR_WAVES, c0slot = 256, ROWP = 272;
    const int lane = threadIdx.x & 63;
    const int wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    const int H = lane >> 5, h = lane & 31;
    const int total = a.n_inst * a.sblocks;
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);
    if (logical >= total) return;
    const int inst = logical / a.sblocks, sb = logical - inst * a.sblocks;
    const HadiInstPar ip = a.ipar[inst];  // (requested here, consumed behind the row fetches: hadi_pass_a_strip)
    const int nrows = a.L.nrows;
    double *ring = smem + (size_t)wave * NS * HADI_PAIR_SLOT;
    double *coef = smem + (size_t)NWV * NS * HADI_PAIR_SLOT;
    double *payl = coef + 4 * 256;
    double *hist = payl + (AMER == 2 ? ROWP : 0) + (size_t)(wave * 2 + H) * 4;  // this half's last four i = 0 values
    // the two strips of this wavefront: 2 (sb NWV + wave) and the next one; the second may be shorter or empty
    const int sA = 2 * (sb * NWV + wave);
    const int j0A = sA * a.RS, j0B = j0A + a.RS;
    const int cntA = HADI_UNIFORM(j0A < nrows ? ((j0A + a.RS < nrows ? j0A + a.RS : nrows) - j0A) : 0);
    const int cntB = HADI_UNIFORM(j0B < nrows ? ((j0B + a.RS < nrows ? j0B + a.RS : nrows) - j0B) : 0);
    const int j0 = H ? j0B : j0A, cnt = H ? cntB : cntA;
    // Strip A comes DOWN, strip B goes UP: the two strips of a wavefront start on either side of their common boundary b = j0B
    // (A on row b - 1, B on row b).  The rows one strip has behind it at the start are then exactly the rows the other one
    // starts on and has one ahead -- they are taken from the other half's ring slots, and the first row from the strip's own:
    // the prologue fetches 3 (4) rows per strip by LDS-DMA and nothing else (before round 4, both strips walking the same
    // way: 3 (4) + 3 register rows, 12 rows per wavefront where 6 are distinct -- the launch's fill burst, all wavefronts at
    // once, is what this pass loses most to).  Strip A of this wavefront and strip B of the one before it END on neighbouring
    // rows at about the same time, as the alternating directions of hadi_pass_a_strip do.
    const int dir = H ? 1 : -1;
    const int js = H ? j0B : j0A + cntA - 1;

    HadiStripCtxT<double> c;
    c.lane = lane; c.rowp = ROWP; c.coef = coef; c.half = 0; c.xch = nullptr; c.err = a.err; c.debug = a.debug;
    const double *__restrict__ Ub = a.U + (size_t)inst * a.L.inst_stride;
    double *__restrict__ Yb = a.Y + (size_t)inst * a.L.inst_stride;
    const double *__restrict__ Lb = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride : nullptr;
    c.Yi = Yb; c.Li = Lb;
    c.b2r = a.b2row + (size_t)inst * ROWP;
    c.inv_dt = 0.0; c.m1_lane = -1; c.m1_r = -1;
    if constexpr (AMER == 2) {
        c.m1_lane = (a.L.m1 - 1) >> 3;
        c.m1_r = (a.L.m1 - 1) & 7;
    }
    // rows of this lane's half, clamped to the grid (out-of-range rows only meet zero weights; a finished half stores nothing)
    auto grow = [&](int jj) { return Ub + (size_t)(jj < 0 ? 0 : (jj >= nrows ? nrows - 1 : jj)) * ROWP; };
    auto slot = [&](int q) { return ring + (size_t)(((q % NS) + NS) % NS) * HADI_PAIR_SLOT; };  // q = step index of the row (any sign)
    const int chunk_off = (h >> 4) * 128 + H * 64 + (h & 15) * 4;  // this lane's first chunk inside a slot (doubles)
    int offA, offB, payA, payB;                                    // ... and where its two pieces sit in the swizzled images
    hadi_pair_offs(h, chunk_off, offA, offB);
    hadi_pair_offs(h, 4 * h, payA, payB);
    const int c0_off = 512 + 16 * H;
    const double *__restrict__ rtab = a.rowc + (size_t)inst * nrows * HADI_RC + HADI_SRC0;
    auto clampj = [&](int jj) { return jj < 0 ? 0 : (jj >= nrows ? nrows - 1 : jj); };
    const int jsA = j0A + cntA - 1, jsB = j0B;  // (wave-uniform)

    // ---- prologue (memory round trips first, then the shared copies and the block's only barrier: hadi_pass_a_strip) ----
    // step index t <-> row js + dir t; the ring slot of a row is its step index mod NS
    double um2[8], um1[8], u0[8];
    // aft[k] = vector-memory instructions issued after the DMA of the row 2 + k ahead (hadi_pass_a_strip)
    constexpr int NA = D - 2;
    int aft[NA];
#pragma unroll
    for (int k = 0; k < NA; k++) aft[k] = 0;
    if (cntA > 0) {
        // the first row too (slot 0; European: the loop's first fetch reuses that slot once the row sits in registers)
        hadi_pair_fetch(grow(js), slot(0), lane);
        hadi_pair_fetch(grow(js + dir), slot(1), lane);
        hadi_pair_fetch(grow(js + 2 * dir), slot(2), lane);
#pragma unroll
        for (int q = 3; q < D; q++) {
            hadi_pair_fetch(grow(js + q * dir), slot(q), lane);
#pragma unroll
            for (int k = 0; k < NA; k++)
                if (k + 2 < q) aft[k] += HADI_PAIR_DMA;
        }
        if (h == 0) {  // the i = 0 values of the rows js - 2 .. js + 1 (steps -2 .. 1)
#pragma unroll
            for (int q = -2; q <= 1; q++) hist[(q + 4) & 3] = grow(js + q * dir)[c0slot];
        }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) um2[r] = um1[r] = u0[r] = 0.0;
    double sc_tmp[4];  // (the shared arrays' global loads before the parameter block is consumed: hadi_pass_a_strip)
    {
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 256;
#pragma unroll
        for (int q = 0; q < 4; q++) sc_tmp[q] = sc[threadIdx.x + q * 64 * NWV];
    }
    // ---- the instance's parameters are consumed here, behind the row fetches ----
    if (n > ip.N) {  // (block-uniform: multi-maturity batches)
#if !defined(HADI_EMU)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of this wavefront may outlive it
#endif
        return;
    }
    c.dt = hadi_uniform_d(ip.dt); c.thdt = hadi_uniform_d(ip.thdt);
    c.c1 = hadi_uniform_d(1.0 + ip.thdt * ip.half_rd);
    c.kap = hadi_uniform_d((ip.dt - ip.thdt) / ip.thdt);
    c.hr0 = hadi_uniform_d(ip.hr0); c.inv0 = hadi_uniform_d(1.0 / (1.0 + ip.thdt * ip.hr0));
    c.e_nm1 = hadi_uniform_d(exp(ip.bc_rate * ip.dt * (n - 1)));  // device_solver.hpp:238
    c.e_n = hadi_uniform_d(exp(ip.bc_rate * ip.dt * n));          // device_solver.hpp:246
    if constexpr (AMER == 2) c.inv_dt = hadi_uniform_d(1.0 / ip.dt);
    {
        const double mq = -(ip.thdt * ip.q);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = threadIdx.x + q * 64 * NWV;
            coef[hadi_pair_swz(e)] = (e < 2 * 256) ? mq * sc_tmp[q] : sc_tmp[q];
        }
    }
    if constexpr (AMER == 2) {
        const double *__restrict__ pg = a.U0 + (size_t)inst * a.L.inst_stride;
        for (int e = threadIdx.x; e < ROWP; e += 64 * NWV) payl[e < 256 ? hadi_pair_swz(e) : e] = pg[e];
    }
    __syncthreads();
    if (cntA == 0) return;
    {   // the first row from this half's slot 0; the rows behind from the OTHER half's slots: its first row is this strip's
        // row behind by one, its row one ahead this strip's row behind by two (rows outside the grid are clamped on both
        // sides alike and only ever meet zero weights).  Every DMA of the prologue has landed: the coefficient copy above
        // consumed loads that were issued behind them (the wait is spelled out all the same).
        hadi_wait_vmcnt(0);
        hadi_wave_rendezvous();
        // (the other half's chunk: the half index is bit 6 of the offsets)
        hadi_pair_get_lds(slot(0), offA, offB, 256, u0);
        hadi_pair_get_lds(slot(0), offA ^ 64, offB ^ 64, 256, um1);
        hadi_pair_get_lds(slot(1), offA ^ 64, offB ^ 64, 256, um2);
#if !defined(HADI_EMU)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // retired before the loop's first fetch reuses slot 0
#endif
    }
    if constexpr (AMER == 2) {  // U = max(P, U_0) on the rows behind
        double pay[8];
        hadi_pair_get_lds(payl, payA, payB, 128, pay);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            um2[r] = fmax(um2[r], pay[r]);
            um1[r] = fmax(um1[r], pay[r]);
        }
    }
#if !defined(HADI_EMU)
#pragma unroll
    for (int r = 0; r < 8; r++) asm volatile("" : "+v"(um2[r]), "+v"(um1[r]), "+v"(u0[r]));
#endif
    hadi_wave_rendezvous();

    for (int t = 0; t < cntA; t++) {
        const int j = js + dir * t;      // this half's row (meaningless once t >= cnt: clamped)
        const bool act = t < cnt;
        HadiSRow srA, srB;  // both halves' row-table entries through the scalar cache; they fly during the DMA wait
        hadi_sload_issue(rtab + (size_t)clampj(jsA - t) * HADI_RC, srA);
        hadi_sload_issue(rtab + (size_t)clampj(jsB + t) * HADI_RC, srB);
        hadi_wave_rendezvous();
        int z = 0;
        if (t + D <= cntA + 1) {  // into the slot of the row that has just left the ring
            hadi_pair_fetch(grow(j + D * dir), slot(t + D), lane);
            z = HADI_PAIR_DMA;
        }
        hadi_wait_vmcnt((NA > 0 ? aft[0] : 0) + z);  // the row two ahead has landed
#pragma unroll
        for (int k = 0; k + 1 < NA; k++) aft[k] = aft[k + 1] + z;
        if (NA > 0) aft[NA - 1] = 0;
        hadi_wave_rendezvous();
        double up1[8], up2[8];
        hadi_pair_get_lds(slot(t + 1), offA, offB, 256, up1);
        hadi_pair_get_lds(slot(t + 2), offA, offB, 256, up2);
        const double c0p2r = slot(t + 2)[c0_off];
        const double c0m2r = hist[(t + 2) & 3], c0m1r = hist[(t + 3) & 3], c00r = hist[t & 3], c0p1r = hist[(t + 1) & 3];
        double rvs[HADI_RCL];
        {
            double rtA[HADI_RCL], rtB[HADI_RCL];
            hadi_sload_wait(srA, rtA);
            hadi_sload_wait(srB, rtB);
            {   // strip A descends: its rows behind are j+1, j+2 -- swap the neighbour weights (scalar registers)
                double w;
                w = rtA[RC_WMS - HADI_SRC0]; rtA[RC_WMS - HADI_SRC0] = rtA[RC_WPS - HADI_SRC0]; rtA[RC_WPS - HADI_SRC0] = w;
                w = rtA[RC_L2 - HADI_SRC0]; rtA[RC_L2 - HADI_SRC0] = rtA[RC_U2 - HADI_SRC0]; rtA[RC_U2 - HADI_SRC0] = w;
                w = rtA[RC_L1 - HADI_SRC0]; rtA[RC_L1 - HADI_SRC0] = rtA[RC_U1 - HADI_SRC0]; rtA[RC_U1 - HADI_SRC0] = w;
            }
            // to per-lane registers under the halves' exec masks (RC_LAST is not needed: is_last below)
#pragma unroll
            for (int k = 0; k < HADI_RCL; k++) rvs[k] = 0.0;
            if (H == 0) {
#pragma unroll
                for (int k = 0; k < HADI_RCL; k++)
                    if (k != RC_LAST - HADI_SRC0) rvs[k] = rtA[k];
            } else {
#pragma unroll
                for (int k = 0; k < HADI_RCL; k++)
                    if (k != RC_LAST - HADI_SRC0) rvs[k] = rtB[k];
            }
        }
        double e0m2 = c0m2r, e0m1 = c0m1r, e00 = c00r, e0p1 = c0p1r, e0p2 = c0p2r, lamc0 = 0.0;
        if constexpr (AMER == 2) {
            double pay[8];
            hadi_pair_get_lds(payl, payA, payB, 128, pay);
            const double pay_c0 = payl[c0slot];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                u0[r] = fmax(u0[r], pay[r]);
                up1[r] = fmax(up1[r], pay[r]);
                up2[r] = fmax(up2[r], pay[r]);
            }
            lamc0 = fmax(0.0, (pay_c0 - c00r) * c.inv_dt);
            e0m2 = fmax(c0m2r, pay_c0); e0m1 = fmax(c0m1r, pay_c0); e00 = fmax(c00r, pay_c0);
            e0p1 = fmax(c0p1r, pay_c0); e0p2 = fmax(c0p2r, pay_c0);
        }
        const bool is_last = act && (j == nrows - 1);
        double un[8];
        double *yrow = Yb + (size_t)(act ? j : 0) * ROWP;
        const double *lrow = (AMER == 1) ? Lb + (size_t)(j < 0 ? 0 : (j >= nrows ? nrows - 1 : j)) * ROWP : nullptr;
        // ONE copy of the step, the b2 terms under the per-half predicate `is_last` (+16 registers, ~24 instructions per row).
        // Two copies selected by "does any half sit on the last row" -- the first version -- were laid out by hipcc as "if (x) A;
        // if (!x) B" with everything B's explicit stage reads (the five max'ed rows, both halves' row scalars) kept alive THROUGH
        // A: 244 live registers in A against 146 in B, 44 - 60 of them spilled into the row loop, 2.4x slower than the kernel
        // it was to replace (tools/experiments/README.md).
        hadi_pair_step<AMER, true>(c, h, act, is_last, rvs, um2, um1, u0, up1, up2, e0m2, e0m1, e00, e0p1, e0p2, lamc0,
                                   slot(t), slot(t + 1), offA, offB, un, yrow, lrow);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            um2[r] = um1[r];
            um1[r] = u0[r];
            u0[r] = un[r];
        }
#pragma unroll
        for (int k = 0; k < NA; k++) aft[k] += HADI_PAIR_STORES;
        if (h == 0) hist[(t + 2) & 3] = c0p2r;  // (raw: step t + 1 reads it as c0p1, ... step t + 4 has overwritten it)
#if !defined(HADI_EMU)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ring re-reads are retired before the next step reuses the slot
#endif
        hadi_wave_rendezvous();
    }
}

