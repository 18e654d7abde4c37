// hadi_k_team.h -- instance-resident launch (hadi_team_kernel): the whole time loop of up to 8 large instances, each kept in one XCD's L2.
// Part of libhadi's device code: include through hadi_kernels.h (which fixes the order).
#pragma once

// ------------------------------------------------------------------------------------------------
// Instance-resident execution: the WHOLE time loop of up to 8 large instances in ONE launch (European Douglas sweeps, fp64,
// one wavefront per v-row: 128 < m1 <= 512, m2 <= 263) -- what the reference's team kernel does for every instance
// (device_solver.hpp:83-88, 226-265: all N steps inside one kernel).  The batched path above needs 2 N dependent launches;
// for ONE 512x256 instance each of them is a few microseconds of work on a sliver of the chip behind a ~1.5 us kernel
// boundary, 17.6 ms per 1000 steps.  Here a TEAM of `nb` blocks, all on the same XCD, keeps the instance in that XCD's L2:
//   row phase     the team's 8 nb wavefronts take the v-rows round-robin: five rows of U straight to registers (L1-bypassing
//                 loads), hadi_strip_step, Y stored;
//   team barrier  every wavefront drains its stores (they are in the XCD's L2 then), one lane per block adds to a
//                 monotonic counter in L2 and polls it;
//   column phase  block t of the team takes column tile t: hadi_pb_load / hadi_pb_solve / hadi_pb_store as in hadi_pass_b;
//   team barrier.
// Which XCD a block runs on is READ from the hardware (HW_REG_XCC_ID), not inferred from blockIdx: blocks that read the
// same id share an L2, so the stores one of them has retired are what the L1-bypassing loads of the others return -- no L2
// write-back, no invalidate, which is what makes the barrier cost ~1 us instead of the 4-5 us of a chip-wide one.  Team k
// = the blocks on XCD k, instance k.  Every wait is bounded; a team that does not fill up (the dispatcher owes nobody a
// round-robin placement), a barrier that runs out of polls, or a block that finds itself on another XCD after a barrier
// (wave save / restore by the driver) records HADI_DEVERR_TEAM in the handle's error word, and the host solves the batch
// again on the streaming path.
struct HadiTeamArgs {
    int *form;   // [8] arrival counters, one per XCD (zeroed before the launch)
    int *bar;    // [8] monotonic barrier counters, one per XCD (zeroed before the launch), each on a cache line of its own
    int nb;      // blocks per team
    int N;       // time steps
    // discrete dividends (HADI_DIV; nullptr = none): the host-built table "which dividend does instance k pay at the START of step
    // n" (hadi_dividend_steps), amounts / percentages, the s-grids in natural order
    const int *div_flag;
    int flag_stride;
    const double *div_amounts, *div_pcts, *vec_s;
    unsigned long long *stamps;  // diagnostic build only (HADI_TEAM_STAMPS): [16]
};
#define HADI_DEVERR_TEAM 2  // instance-resident launch: a team did not form, a team barrier timed out, or a block moved
#define HADI_TEAM_POLLS (1 << 18)

// Diagnostic build only (-DHADI_TEAM_STAMPS, tools/team_stamps.py): shader-clock stamps of the phases of time step
// HADI_TEAM_STAMPS, written by wavefront 0 of the team's blocks 0 (a column-phase block) and nb - 1 (a row-phase-only block).
#if defined(HADI_TEAM_STAMPS) && !defined(HADI_EMU)
#define HADI_TSTAMP(k, drain) do { if (n == HADI_TEAM_STAMPS && wave == 0 && (rank == 0 || rank == nb - 1)) { \
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (lane == 0) ta.stamps[(rank == 0 ? 0 : 8) + (k)] = t_; } } while (0)
#else
#define HADI_TSTAMP(k, drain)
#endif

HADI_DEV HADI_FORCEINLINE int hadi_xcc_id() {
#if defined(HADI_EMU)
    return (int)(blockIdx.x & 7);
#else
    return (int)(__builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 7);
#endif
}
// Cross-CU visibility inside a team (HADI_TEAM_COHERENCE): how a wavefront is guaranteed to see what ANOTHER CU of its XCD wrote
// before the team barrier.  The writer side is the same either way -- every wavefront drains its stores (the vector L1 is
// write-through: an acknowledged store is in the XCD's L2) before its block arrives at the barrier.
//   1  reader: agent-coherent loads (cache policy sc1, on top of nt): served by the L2, never by this CU's vector L1 -- the
//      gfx942 / gfx950 memory model's "load atomic monotonic, agent scope", applied to the only loads that cross CUs (the five
//      stencil rows of the row phase, the tile of the column phase).  No invalidate.
//   2  reader: `buffer_inv sc1` behind the barrier's poll (the model's agent-scope acquire), plain nt loads.  Measured 1.7 us
//      per time step slower than (1) on 512x256 (16.2 -> 17.9 ms per 1000 steps, profiles/r04_team_ab.txt): the invalidate also
//      throws out the row tables and the lines the next phase's first loads would have hit.
//   0  nt loads only (round 3): nt is a streaming HINT, not a coherence guarantee.  Kept for A/B timing only.
#ifndef HADI_TEAM_COHERENCE
#define HADI_TEAM_COHERENCE 1
#endif
// this lane's B values of a row-layout global row of the instance behind `ub` (byte offset `row_bytes`, wave-uniform)
template <int B>
HADI_DEV HADI_FORCEINLINE void hadi_get_block_l2(HadiBuf ub, const double *row, unsigned row_bytes, int lane, double (&u)[B]) {
#if !defined(HADI_EMU)
    typedef double hadi_d2 __attribute__((ext_vector_type(2)));
#endif
#pragma unroll
    for (int q = 0; q < B / 2; q++) {
#if defined(HADI_EMU)
        (void)ub; (void)row_bytes;
        u[2 * q] = row[q * 128 + 2 * lane]; u[2 * q + 1] = row[q * 128 + 2 * lane + 1];
#elif HADI_TEAM_COHERENCE == 1
        (void)row;
        hadi_buf_load2_sc1(ub, (unsigned)(q * 128 + 2 * lane) * 8u, row_bytes, u[2 * q], u[2 * q + 1]);
#else
        (void)ub; (void)row_bytes;
        const hadi_d2 t = __builtin_nontemporal_load(reinterpret_cast<const hadi_d2 *>(row + q * 128 + 2 * lane));
        u[2 * q] = t.x; u[2 * q + 1] = t.y;
#endif
    }
}
HADI_DEV HADI_FORCEINLINE double hadi_get_l2(HadiBuf ub, const double *p, unsigned off_bytes) {
#if defined(HADI_EMU)
    (void)ub; (void)off_bytes;
    return *p;
#elif HADI_TEAM_COHERENCE == 1
    (void)p;
    return hadi_buf_load_sc1(ub, 0u, off_bytes);
#else
    (void)ub; (void)off_bytes;
    return __builtin_nontemporal_load(p);
#endif
}
// Barrier of the team's blocks.  `dead` (LDS) is set when the wait ran out of polls or the block has moved; returns false then.
HADI_DEV HADI_FORCEINLINE bool hadi_team_barrier(int *ctr, int target, int xcc_team, int *dead, int *err) {
#if !defined(HADI_EMU)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's stores have been acknowledged by the L2
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
#if defined(HADI_EMU)
        __atomic_fetch_add(ctr, 1, __ATOMIC_SEQ_CST);
        int guard = 0;
        // (host threads: a phase of another block takes tenths of a second here -- back off instead of burning the poll budget)
        while (__atomic_load_n(ctr, __ATOMIC_SEQ_CST) < target && ++guard < HADI_TEAM_POLLS) {
            if (guard < 1024) sched_yield();
            else { struct timespec ts = {0, 100000}; nanosleep(&ts, nullptr); }
        }
#else
        // Release side: every wavefront of the block drained its stores above (the vector L1 is write-through: an
        // acknowledged store IS in this XCD's L2), so the counter update itself can be relaxed.  An agent-scope RELEASE
        // would add `buffer_wbl2 sc1` -- a write-back of the L2's dirty lines, i.e. of the instance the team keeps there on
        // purpose -- for readers that share this very L2 (checked below through HW_REG_XCC_ID).
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int guard = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++guard < HADI_TEAM_POLLS)
            __builtin_amdgcn_s_sleep(2);
        // Acquire side: see HADI_TEAM_COHERENCE above -- agent-coherent (sc1) loads of everything that crosses CUs, or, in the
        // alternative build, an invalidate of this CU's vector L1 right here.
#if HADI_TEAM_COHERENCE == 2
        asm volatile("buffer_inv sc1" ::: "memory");
#endif
#endif
        if (guard >= HADI_TEAM_POLLS || hadi_xcc_id() != xcc_team) {
            hadi_report(err, HADI_DEVERR_TEAM);
            *dead = 1;
        }
    }
    __syncthreads();
    return *dead == 0;
}

template <int B>
__global__ void __launch_bounds__(512, 2) hadi_team_kernel(HadiSweepArgs a, HadiTeamArgs ta) {
    HADI_DYN_SMEM(double, smem);
    constexpr int c0slot = 64 * B;
    const int lane = threadIdx.x & 63;
    const int wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    const int xcc = hadi_xcc_id();
    if (xcc >= a.n_inst) return;  // (block-uniform: a workgroup lives on one XCD)
    // LDS: [4 coefficient arrays of 64 B] [the column pass's reduced system: exchange values Z, selected inverse rows RT, their
    //      product T (hadi_pb_mf_doubles)] [P chunk tables of the column pass] [flags]
    const int P = a.L.P, n4 = 4 * P;
    double *coef = smem;
    double *zsh = coef + 4 * 64 * B;
    double *rtsh = zsh + (size_t)n4 * 64;
    double *tprod = rtsh + (size_t)n4 * hadi_pb_mp(P);
    double *tabl = tprod + (size_t)hadi_pb_mp(P) * 64;  // the column-pass chunk tables, [P][HADI_LC][HADI_PBW]
    // discrete dividends only: the instance's s-grid (natural order) and one row of scratch per wavefront
    double *sgrid = tabl + (size_t)P * HADI_LC * HADI_PBW;
    double *drow = sgrid + (ta.div_flag ? (a.L.m1 + 2) : 0) + (size_t)wave * a.L.rowp;
    int *flags = reinterpret_cast<int *>(sgrid + (ta.div_flag ? (a.L.m1 + 2) + (size_t)8 * a.L.rowp : 0));  // [0] rank of this block in its team, [1] dead
    if (threadIdx.x == 0) {
#if defined(HADI_EMU)
        flags[0] = __atomic_fetch_add(ta.form + xcc, 1, __ATOMIC_SEQ_CST);
#else
        flags[0] = __hip_atomic_fetch_add(ta.form + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        flags[1] = 0;
    }
    __syncthreads();
    const int rank = HADI_UNIFORM(flags[0]);
    const int nb = ta.nb;
    if (rank >= nb) return;  // more blocks of the grid landed on this XCD than the team takes
    if ((a.debug & HADI_DEBUG_TEAM_DESERT) && rank == 1) return;  // (test hook)
    const int inst = xcc;
    const HadiInstPar ip = a.ipar[inst];
    const int nrows = a.L.nrows, rowp = a.L.rowp;
    {   // s-coefficient arrays, the beta pair scaled as in hadi_pass_a_strip; this wavefront's rows of the reduced inverse
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 64 * B;
        const double mq = -(ip.thdt * ip.q);
        for (int e = threadIdx.x; e < 4 * 64 * B; e += 512) coef[e] = (e < 2 * 64 * B) ? mq * sc[e] : sc[e];
        hadi_pb_stage_rt(a.rinv + (size_t)inst * 16 * P * P, P, rtsh, 512);
        const double *__restrict__ pg = a.pb + (size_t)inst * a.L.nrows_pad * HADI_PBW;
        for (int e = threadIdx.x; e < P * HADI_LC * HADI_PBW; e += 512) tabl[e] = pg[e];
        if (ta.div_flag) {
            const double *__restrict__ sg = ta.vec_s + (size_t)inst * (a.L.m1 + 1);
            for (int e = threadIdx.x; e <= a.L.m1; e += 512) sgrid[e] = sg[e];
        }
    }
    __syncthreads();

    HadiStripCtxT<double> c;
    c.lane = lane; c.rowp = rowp; c.coef = coef; c.half = 0; c.xch = nullptr; c.err = a.err; c.debug = 0;
    c.dt = hadi_uniform_d(ip.dt); c.thdt = hadi_uniform_d(ip.thdt);
    c.c1 = hadi_uniform_d(1.0 + ip.thdt * ip.half_rd);
    c.kap = hadi_uniform_d((ip.dt - ip.thdt) / ip.thdt);
    c.hr0 = hadi_uniform_d(ip.hr0); c.inv0 = hadi_uniform_d(1.0 / (1.0 + ip.thdt * ip.hr0));
    double *const Ui = a.U + (size_t)inst * a.L.inst_stride;
    c.Yi = a.Y + (size_t)inst * a.L.inst_stride;
    c.Li = nullptr; c.R1i = nullptr; c.C2i = nullptr;
    c.b2r = a.b2row + (size_t)inst * rowp;
    c.inv_dt = 0.0; c.m1_lane = -1; c.m1_r = -1;

    HadiPassBCtx cb;
    cb.lane = lane; cb.wave = wave; cb.P = P; cb.zsh = zsh; cb.Ri = nullptr; cb.RT = rtsh; cb.Tsh = tprod;
    cb.nrows = nrows; cb.rowp = rowp; cb.ja = wave * HADI_LC;
    cb.Yi = c.Yi; cb.Ui = Ui;
    cb.Yb = hadi_make_buf(c.Yi, (size_t)a.L.inst_stride * sizeof(double));
    cb.Ub = hadi_make_buf(Ui, (size_t)a.L.inst_stride * sizeof(double));
    cb.Li = nullptr; cb.Lb = hadi_make_buf(nullptr, 0); cb.P0i = nullptr; cb.pay1d = 0; cb.inv_dt = 0.0;
    cb.american = 0; cb.debug = 0; cb.pos_m1 = a.pos_m1; cb.dt = ip.dt;
    cb.tabl = tabl + (size_t)wave * HADI_LC * HADI_PBW;
    if (wave < P) hadi_pb_load_table(cb, a.pb + ((size_t)inst * a.L.nrows_pad + cb.ja) * HADI_PBW);

    const int wt = rank * 8 + wave, nwt = nb * 8;  // this wavefront's number in the team
    int *const bar = ta.bar + 32 * xcc;
    int arrivals = 0;
    const int N = ip.N < ta.N ? ip.N : ta.N;
    // boundary time factors e_n = exp(bc_rate dt n) (device_solver.hpp:238,246): one exp per step (e_{n-1} is last step's
    // e_n, the same bits), none for the call with r_f = 0 (exp(0) = 1 exactly)
    const bool unit_e = (ip.bc_rate == 0.0);
    double e_cur = 1.0;  // exp(bc_rate dt 0)
    for (int n = 1; n <= N; n++) {
        c.e_nm1 = hadi_uniform_d(e_cur);
        if (!unit_e) e_cur = exp(ip.bc_rate * ip.dt * n);
        c.e_n = hadi_uniform_d(e_cur);
        HADI_TSTAMP(0, false);
        // ---- discrete dividend at the START of the step (device_solver.hpp:426-517; hadi_dividend_kernel) -----------------------
        // The jump acts on every v-row by itself: a wavefront copies its row of U (written by other CUs in the column phase:
        // coherent loads) to its LDS scratch row, then every lane rebuilds its nodes by linear interpolation at the ex-dividend
        // spot and stores them in place.  One team barrier more on the <= num_dividends steps that pay.
        {
            const int dv = ta.div_flag ? ta.div_flag[(size_t)inst * ta.flag_stride + n - 1] : -1;  // (wave-uniform)
            if (dv >= 0) {
                const double amount = ta.div_amounts[dv], pct = ta.div_pcts[dv];
                const int m1 = a.L.m1;
                for (int j = wt; j < nrows; j += nwt) {
                    double *r0 = Ui + (size_t)j * rowp;
                    double rv[B];
                    hadi_get_block_l2<B>(cb.Ub, r0, (unsigned)j * (unsigned)rowp * 8u, lane, rv);
                    const double c00 = hadi_get_l2(cb.Ub, r0 + c0slot, ((unsigned)j * (unsigned)rowp + (unsigned)c0slot) * 8u);
                    hadi_wave_rendezvous();
                    hadi_put_block<B, 1>(drow, 0, lane, rv);
                    if (lane == 0) drow[c0slot] = c00;
                    hadi_wave_rendezvous();  // (one wavefront: LDS writes are visible to its own later reads; the emulator needs the barrier)
#if !defined(HADI_EMU)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    auto jump = [&](int i) -> double {
                        const double new_s = sgrid[i] * (1.0 - pct) - amount;
                        double out = ip.put ? drow[c0slot] : 0.0;  // ex-dividend spot <= 0: a call is worth 0, a put its s = 0 value
                        if (new_s > 0) {
                            int lo = 0, hi = m1 + 1;  // first k in [0, m1] with s[k] > new_s, 0 if none
                            while (lo < hi) {
                                const int mid = (lo + hi) >> 1;
                                if (sgrid[mid] > new_s) hi = mid;
                                else lo = mid + 1;
                            }
                            const int idx = (lo <= m1) ? lo : 0;
                            if (idx > 0) {
                                const double s_low = sgrid[idx - 1], s_high = sgrid[idx];
                                const double weight = (new_s - s_low) / (s_high - s_low);
                                out = (1.0 - weight) * drow[hadi_pos(B, 1, idx - 1)] + weight * drow[hadi_pos(B, 1, idx)];
                            } else {
                                out = drow[c0slot];
                            }
                        }
                        return out;
                    };
                    double nv[B];
#pragma unroll
                    for (int r = 0; r < B; r++) {
                        const int i = 1 + B * lane + r;
                        nv[r] = (i <= m1) ? jump(i) : 0.0;  // (slots beyond m1 are zero pads)
                    }
                    const double n0 = jump(0);
                    hadi_put_block<B, 1>(r0, 0, lane, nv);
                    if (lane == 0) r0[c0slot] = n0;
                }
                arrivals += nb;
                if (!hadi_team_barrier(bar, arrivals, xcc, flags + 1, a.err)) return;
            }
        }
        // ---- row phase ---------------------------------------------------------------------------------------------
        for (int j = (a.debug & HADI_DEBUG_TEAM_NO_ROWS) ? nrows : wt; j < nrows; j += nwt) {
            HadiSRow srow;
            hadi_sload_issue(a.rowc + ((size_t)inst * nrows + j) * HADI_RC + HADI_SRC0, srow);
            double um2[B], um1[B], u0[B], up1[B], up2[B], un[B], praw[B];
#pragma unroll
            for (int r = 0; r < B; r++) um2[r] = um1[r] = up1[r] = up2[r] = praw[r] = 0.0;
            double c0m2 = 0.0, c0m1 = 0.0, c0p1 = 0.0, c0p2 = 0.0;
            const double *r0 = Ui + (size_t)j * rowp;
            const unsigned rb = (unsigned)rowp * 8u, o0 = (unsigned)j * rb, oc = (unsigned)c0slot * 8u;  // (byte offsets inside the instance)
            if (j >= 2) { hadi_get_block_l2<B>(cb.Ub, r0 - 2 * rowp, o0 - 2 * rb, lane, um2); c0m2 = hadi_get_l2(cb.Ub, r0 - 2 * rowp + c0slot, o0 - 2 * rb + oc); }
            if (j >= 1) { hadi_get_block_l2<B>(cb.Ub, r0 - rowp, o0 - rb, lane, um1); c0m1 = hadi_get_l2(cb.Ub, r0 - rowp + c0slot, o0 - rb + oc); }
            hadi_get_block_l2<B>(cb.Ub, r0, o0, lane, u0);
            const double c00 = hadi_get_l2(cb.Ub, r0 + c0slot, o0 + oc);
            if (j + 1 < nrows) { hadi_get_block_l2<B>(cb.Ub, r0 + rowp, o0 + rb, lane, up1); c0p1 = hadi_get_l2(cb.Ub, r0 + rowp + c0slot, o0 + rb + oc); }
            if (j + 2 < nrows) { hadi_get_block_l2<B>(cb.Ub, r0 + 2 * rowp, o0 + 2 * rb, lane, up2); c0p2 = hadi_get_l2(cb.Ub, r0 + 2 * rowp + c0slot, o0 + 2 * rb + oc); }
            double rt[HADI_RCL];
            hadi_sload_wait(srow, rt);
            hadi_wave_rendezvous();
            if (j == nrows - 1)
                hadi_strip_step<B, 0, true, double, 1>(c, j, rt, um2, um1, u0, up1, up2, hadi_uniform_d(c0m2), hadi_uniform_d(c0m1), hadi_uniform_d(c00),
                                                       hadi_uniform_d(c0p1), hadi_uniform_d(c0p2), praw, 0.0, coef, un);
            else
                hadi_strip_step<B, 0, false, double, 1>(c, j, rt, um2, um1, u0, up1, up2, hadi_uniform_d(c0m2), hadi_uniform_d(c0m1), hadi_uniform_d(c00),
                                                        hadi_uniform_d(c0p1), hadi_uniform_d(c0p2), praw, 0.0, coef, un);
        }
        HADI_TSTAMP(1, false);
        arrivals += nb;
        if (!(a.debug & HADI_DEBUG_TEAM_NO_BARRIER) && !hadi_team_barrier(bar, arrivals, xcc, flags + 1, a.err)) return;
        HADI_TSTAMP(2, false);
        // ---- column phase: tile t on block t of the team ---------------------------------------------------------------
        for (int t = (a.debug & HADI_DEBUG_TEAM_NO_COLS) ? a.ctiles : rank; t < a.ctiles; t += nb) {
            if (wave < P) {
                double y[HADI_LC];
                hadi_pb_load<double, HADI_TEAM_COHERENCE == 1>(cb, t, y);
                HADI_TSTAMP(3, true);   // (diagnostic build: waits for the loads)
                hadi_pb_solve<false, false, true>(cb, 0, y, 0);  // (reduced system on the matrix core: two block barriers inside)
                HADI_TSTAMP(4, false);
                hadi_pb_store<0, false, double>(cb, t, y);
                HADI_TSTAMP(5, false);
            } else if (P > 1) {
                __syncthreads();  // (the two barriers inside hadi_pb_solve: exchange values written, product written)
                __syncthreads();
            }
        }
        arrivals += nb;
        if (!(a.debug & HADI_DEBUG_TEAM_NO_BARRIER) && !hadi_team_barrier(bar, arrivals, xcc, flags + 1, a.err)) return;
        HADI_TSTAMP(6, false);
    }
}

