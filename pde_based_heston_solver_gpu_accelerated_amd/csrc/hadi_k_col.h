// hadi_k_col.h -- column pass (hadi_pb_*: chunk sweeps, SPIKE reduced system on the matrix core, projection; hadi_pass_b, hadi_pass_b1, hadi_pass_b2).
// Part of libhadi's device code: include through hadi_kernels.h (which fixes the order).
#pragma once

// ------------------------------------------------------------------------------------------------
// pass B.  Block = P wavefronts (P*64 threads); wavefront p owns v-rows [p*HADI_LC, (p+1)*HADI_LC) of its
// instance (rows past nrows are identity padding, so there are no tail branches) and
// keeps one 64-column tile of them in registers.  A block walks over `btpw` consecutive column tiles
// with two register buffers: the loads of tile t+1 are in flight while tile t is solved and stored, so
// the memory pipe stays busy through the dependent forward/backward chains.
struct HadiPassBCtx {
    const double *Yi;   // instance base of Y
    double *Ui;         // instance base of U
    HadiBuf Yb, Ub;     // the same two as buffer resources (uniform)
    HadiBuf Lb;         // lambda_bar as a buffer resource (American, explicit pair)
    double *Li;         // instance base of lambda_bar (American)
    const double *P0i;  // instance base of the payoff (American)
    int pay1d;          // the payoff does not depend on v: one load per column instead of one per node
    double inv_dt;      // 1/dt (P representation)
    double tab[5];      // this chunk's table (33 rows x 9 scalars) spread over the lanes: lane l holds entries l + 64 q
    const double *tabl; // hadi_pb_solve<true>: the chunk's table rows in LDS, [HADI_LC][HADI_PBW] (instance-resident kernel)
    const double *Ri;   // this wavefront's four rows of the reduced inverse in LDS, [4P][4]: for column m the
                        // coefficients of (left-neighbour last two, right-neighbour first two)
    double *zsh;        // LDS exchange, 2 buffers of P*4*64 (ONE buffer with the matrix-core reduced system, below)
    const double *RT;   // hadi_pb_solve<.., MF>: the selected rows of the reduced inverse in LDS, TRANSPOSED: RT[k][r], r = 4 w + q the
                        // q-th coefficient row of wavefront w (0, 1: its left neighbour's last two unknowns, 2, 3: its right
                        // neighbour's first two), pitch MP = 16 ceil(P / 4), zero beyond r = 4 P
    double *Tsh;        // ... and its result T = R Z in LDS, [MP][64]
    int lane, wave, P, ja, rowp, american, pos_m1;
    int nrows;          // real v-rows (m2 + 1); rows nrows .. P*HADI_LC-1 are identity padding
    double dt;
    int debug;          // HadiSweepArgs.debug (timing diagnostics only)
    HADI_STAMP_ACC
};

// Scalar m of chunk row k of the column-pass table.  The table is identical for all 64 columns; reading it from LDS costs
// a full-width LDS return per row and phase (a broadcast ds_read_b128 still moves 1 KiB to the VGPRs): ~280 reads per
// tile and wavefront, which kept the LDS pipe of the CU busy for a third of the tile time.  Spread over the lanes of 5
// register pairs and fetched with v_readlane (static lane index, result in SGPRs, used as an FMA operand) it costs
// only VALU slots, of which this kernel has plenty.
#if defined(HADI_EMU)  // same register / lane mapping, read from the image the lanes published after loading the table
#define HADI_PB_T(c, k, m) (emu::t_wave->pub[((k) * 9 + (m)) >> 6][((k) * 9 + (m)) & 63])
#else
#define HADI_PB_T(c, k, m) hadi_read_lane((c).tab[((k) * 9 + (m)) >> 6], ((k) * 9 + (m)) & 63)
#endif

HADI_DEV HADI_FORCEINLINE void hadi_pb_load_table(HadiPassBCtx &c, const double *__restrict__ pbg) {
    static_assert(HADI_LC * 9 <= 5 * 64, "table does not fit 5 registers per lane");
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int e = c.lane + 64 * q;
        const int k = e / 9, m = e - k * 9;
        // (image positions 0, 1 hold the SCALED forward multipliers PB_LQ, PB_L2Q: the kernels never use the raw ones)
        c.tab[q] = (k < HADI_LC) ? pbg[(size_t)k * HADI_PBW + (m == PB_L ? PB_LQ : m == PB_L2 ? PB_L2Q : m)] : 0.0;
#if defined(HADI_EMU)
        emu::t_wave->pub[q][c.lane] = c.tab[q];
#endif
    }
    hadi_wave_rendezvous();
}

// Storage row of chunk row k.  Identity padding rows (>= nrows; only the last chunk has any) are all mapped to the FIRST
// padding row: they hold zeros in Y (the row pass never writes them) and receive zeros in U, so one cached row serves
// every padded load and store and the padding costs no HBM traffic (scalar min + multiply per row, no VALU).
HADI_DEV HADI_FORCEINLINE unsigned hadi_pb_row(const HadiPassBCtx &c, int k) {
    const int r = c.ja + k;
    return (unsigned)(r < c.nrows ? r : c.nrows);
}

template <class T = double, bool SC1 = false>
HADI_DEV HADI_FORCEINLINE void hadi_pb_load(const HadiPassBCtx &c, int ctile, double (&y)[HADI_LC]) {
    constexpr unsigned ES = (unsigned)sizeof(T);
    const int col = ctile * 64 + c.lane;
    const int colc = col < c.rowp ? col : c.rowp - 1;  // lanes past the pitch read a valid address, never store
    const unsigned voff = (unsigned)colc * ES;
    const unsigned rstride = (unsigned)c.rowp * ES;
#pragma unroll
    for (int k = 0; k < HADI_LC; k++) {
        if constexpr (SC1) y[k] = hadi_buf_load_sc1(c.Yb, voff, hadi_pb_row(c, k) * rstride);  // (agent-coherent: the team kernel)
        else y[k] = hadi_buf_load_t<T>(c.Yb, voff, hadi_pb_row(c, k) * rstride);
    }
}

// D(16 x 16) += A(16 x 4) B(4 x 16) on the matrix core, fp64 (v_mfma_f64_16x16x4_f64).  Operand mapping, verified on gfx950 by
// tools/mfma_probe.hip: lane 16 k + i holds A[i][k], lane 16 k + j holds B[k][j]; register r of lane 16 q + j holds D[4 r + q][j].
HADI_DEV HADI_FORCEINLINE void hadi_mfma_f64_16x16x4(double a, double b, double (&acc)[4]) {
#if defined(HADI_EMU)
    const int q = emu::t_lane >> 4, j = emu::t_lane & 15;
    double av[4][4], bv[4];
    for (int k = 0; k < 4; k++) {
        bv[k] = __shfl(b, 16 * k + j);
        for (int r = 0; r < 4; r++) av[r][k] = __shfl(a, 16 * k + 4 * r + q);
    }
    for (int r = 0; r < 4; r++)
        for (int k = 0; k < 4; k++) acc[r] = fma(av[r][k], bv[k], acc[r]);
#else
    typedef double hadi_d4 __attribute__((ext_vector_type(4)));
    hadi_d4 c = {acc[0], acc[1], acc[2], acc[3]};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    acc[0] = c[0]; acc[1] = c[1]; acc[2] = c[2]; acc[3] = c[3];
#endif
}
// Stages RT for instance `inst` (whole block; the caller's barrier follows).  R^-1 is the dense inverse of the 4 P x 4 P SPIKE
// reduced system (hadi_setup_instance); wavefront w needs the rows of its left neighbour's last two and its right neighbour's
// first two unknowns (spikes are zero where there is no neighbour, so any row will do there).
HADI_DEV HADI_FORCEINLINE void hadi_pb_stage_rt(const double *__restrict__ Rg, int P, double *__restrict__ RT, int nthreads) {
    const int n4 = 4 * P, MP = hadi_pb_mp(P);
    for (int e = threadIdx.x; e < n4 * MP; e += nthreads) {
        const int k = e / MP, r = e - k * MP, w = r >> 2, q = r & 3;
        double v = 0.0;
        if (w < P) {
            const int rl0 = (w > 0) ? 4 * (w - 1) + 2 : 0, rr0 = (w < P - 1) ? 4 * (w + 1) : 0;
            v = Rg[(size_t)((q < 2) ? rl0 + q : rr0 + (q - 2)) * n4 + k];
        }
        RT[e] = v;
    }
}

// The column pass's LDS (whole block; the caller's barrier follows).  Returns the first free double behind it.
//   MF:      [Z: 4 P x 64] [RT: 4 P x MP] [T: MP x 64]
//   else:    [Z: 1 or 2 buffers of 4 P x 64] [each wavefront's four rows of the reduced inverse: P x 4 x 4 P]
template <bool MF>
HADI_DEV HADI_FORCEINLINE double *hadi_pb_setup_lds(HadiPassBCtx &c, double *smem, const double *__restrict__ Rg, int zbuffers) {
    const int P = c.P, n4 = 4 * P;
    c.zsh = smem;
    c.RT = nullptr; c.Tsh = nullptr; c.Ri = nullptr;
    if constexpr (MF) {
        double *RT = smem + (size_t)n4 * 64;
        hadi_pb_stage_rt(Rg, P, RT, 64 * P);
        c.RT = RT;
        c.Tsh = RT + (size_t)n4 * hadi_pb_mp(P);
        return c.Tsh + (size_t)hadi_pb_mp(P) * 64;
    } else {
        double *tsh = smem + (size_t)P * zbuffers * 4 * 64;
        double *__restrict__ rw = tsh + (size_t)c.wave * 4 * n4;
        const int rl0 = (c.wave > 0) ? 4 * (c.wave - 1) + 2 : 0, rr0 = (c.wave < P - 1) ? 4 * (c.wave + 1) : 0;
        for (int e = c.lane; e < 4 * n4; e += 64) {
            const int m = e >> 2, q = e & 3;
            rw[e] = Rg[(size_t)((q < 2) ? rl0 + q : rr0 + (q - 2)) * n4 + m];
        }
        c.Ri = rw;
        return tsh + (size_t)16 * P * P;
    }
}

// Chunk-local solve + interface exchange + spike correction of one 64-column tile held in y (no memory traffic).
// LDSTAB: the table scalars come from an LDS copy (broadcast reads) instead of v_readlane on the register image.  The
// streaming kernels stream tiles through eight wavefronts and are short of LDS bandwidth, not of VALU slots: readlanes there.
// The instance-resident kernel solves ONE tile per step and waits for it: there the 594 readlanes per tile (two VALU slots
// per scalar) are a third of the phase's instruction chain, and 165 broadcast reads replace them.
// ONEBUF: ONE exchange buffer instead of two alternating ones -- a second block-wide barrier in front of the exchange
// write keeps a fast wavefront from overwriting values a slow one still reads (hadi_pass_b2: the LDS of the second buffer
// holds prefetched rows there).
#ifndef HADI_PB_MF
#define HADI_PB_MF 1  // the streaming column kernels run the reduced system on the matrix core (0: the broadcast-operand FMA loop, for A/B builds)
#endif
// MF: the reduced system t = R^-1 z on the MATRIX CORE.  Every wavefront needs four rows of T = R Z (its own four coefficient
// rows times the 4 P x 64 exchange values of the tile's 64 columns): together the block computes a dense (4 P) x (4 P) x 64
// product per tile.  As 4 x 4 P broadcast-operand FMAs per lane that product was two thirds of the whole solve -- every FMA
// pulled a wave-uniform coefficient through the LDS return path (two broadcast ds_read_b128 per exchange row and wavefront:
// 2.5 MB per 16-chunk tile, ~8 us of LDS pipe per tile against ~1.7 us of arithmetic; gpurun_out/r04g/colpass_ab.txt: the
// 1024x512 column pass 0.136 -> 0.112 ms per launch with the loop removed).  Here wavefront w computes the 16 x 16 block
// (w / 4, w % 4) of T with P v_mfma_f64_16x16x4_f64 (operands: one double per lane each, conflict-free ds_read_b64), writes it
// to LDS, and after a second barrier every wavefront picks its four rows.  ONE exchange buffer: the two barriers of a tile
// order every reuse.  The stencil sweep itself stays on the vector units; this is the one dense contraction of the scheme.
template <bool LDSTAB = false, bool ONEBUF = false, bool MF = false>
HADI_DEV HADI_FORCEINLINE void hadi_pb_solve(const HadiPassBCtx &c, int parity, double (&y)[HADI_LC], int younger = 0) {
#undef HADI_PB_T
#if defined(HADI_EMU)
#define HADI_PB_T(c, k, m) (LDSTAB ? (c).tabl[(k) * HADI_PBW + (m)] : emu::t_wave->pub[((k) * 9 + (m)) >> 6][((k) * 9 + (m)) & 63])
#else
#define HADI_PB_T(c, k, m) (LDSTAB ? (c).tabl[(k) * HADI_PBW + (m)] : hadi_read_lane((c).tab[((k) * 9 + (m)) >> 6], ((k) * 9 + (m)) & 63))
#endif
    HADI_STAMP_DECL(c.stamp_acc_)
    HADI_STAMPB_WAIT(younger);
    HADI_STAMPB(16);  // this tile's loads have landed
    // forward elimination with the chunk-local factorisation
    {
        double ym1 = 0.0, ym2 = 0.0;
#pragma unroll
        for (int k = 0; k < HADI_LC; k++) {
            // (scaled multipliers: one operation on the dependent chain; LDSTAB reads table slots, the register image keeps the
            // scaled pair in positions PB_L, PB_L2)
            const double lq = LDSTAB ? c.tabl[k * HADI_PBW + PB_LQ] : HADI_PB_T(c, k, PB_L);
            const double l2q = LDSTAB ? c.tabl[k * HADI_PBW + PB_L2Q] : HADI_PB_T(c, k, PB_L2);
            const double yk = fma(-lq, ym1, fma(-l2q, ym2, y[k] * HADI_PB_T(c, k, PB_Q)));
            y[k] = yk;
            ym2 = ym1;
            ym1 = yk;
        }
    }
    HADI_STAMPB(17);  // forward
    // back substitution
    {
        double xp1 = 0.0, xp2 = 0.0;
#pragma unroll
        for (int k = HADI_LC - 1; k >= 0; k--) {
            const double xk = fma(-HADI_PB_T(c, k, PB_C), xp1, fma(-HADI_PB_T(c, k, PB_C2), xp2, y[k]));
            y[k] = xk;
            xp2 = xp1;
            xp1 = xk;
        }
    }
    HADI_STAMPB(18);  // backward
    const int P = c.P;
    if (P > 1) {
        // interface exchange.  Two LDS buffers alternate by tile parity, so one barrier per tile is
        // enough: a wave can only overwrite buffer b two tiles later, after every wave has passed the
        // barrier of the tile in between, i.e. has finished reading b.
        double *__restrict__ z = c.zsh + (size_t)((ONEBUF || MF) ? 0 : parity) * P * 4 * 64;
        if constexpr (ONEBUF && !MF) __syncthreads();  // every wavefront has read the previous tile's exchange values
        z[(c.wave * 4 + 0) * 64 + c.lane] = y[0];
        z[(c.wave * 4 + 1) * 64 + c.lane] = y[1];
        z[(c.wave * 4 + 2) * 64 + c.lane] = y[HADI_LC - 2];
        z[(c.wave * 4 + 3) * 64 + c.lane] = y[HADI_LC - 1];
        __syncthreads();
        HADI_STAMPB(19);  // exchange + barrier
        // t = Rinv z : this chunk needs the previous chunk's last two and the next chunk's first two
        const int n4 = 4 * P;
        double tl0 = 0.0, tl1 = 0.0, tr0 = 0.0, tr1 = 0.0;
        if constexpr (MF) {
            const int MP = hadi_pb_mp(P), nblk = MP >> 2;  // (MP / 16 row blocks x 4 column blocks)
            const int kq = c.lane >> 4, ij = c.lane & 15;
            for (int blk = c.wave; blk < ((c.debug & HADI_DEBUG_COL_NO_REDUCED) ? 0 : nblk); blk += P) {
                const int bi = blk >> 2, bj = blk & 3;
                const double *__restrict__ ap = c.RT + kq * MP + 16 * bi + ij;
                const double *__restrict__ bp = z + kq * 64 + 16 * bj + ij;
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                for (int sK = 0; sK < P; sK++) hadi_mfma_f64_16x16x4(ap[(size_t)sK * 4 * MP], bp[(size_t)sK * 4 * 64], acc);
#pragma unroll
                for (int r = 0; r < 4; r++) c.Tsh[(16 * bi + 4 * r + kq) * 64 + 16 * bj + ij] = acc[r];
            }
            __syncthreads();
            tl0 = c.Tsh[(4 * c.wave + 0) * 64 + c.lane];
            tl1 = c.Tsh[(4 * c.wave + 1) * 64 + c.lane];
            tr0 = c.Tsh[(4 * c.wave + 2) * 64 + c.lane];
            tr1 = c.Tsh[(4 * c.wave + 3) * 64 + c.lane];
        } else {
            const double *__restrict__ Ri = c.Ri;
#pragma unroll 8
            for (int m = 0; m < ((c.debug & HADI_DEBUG_COL_NO_REDUCED) ? 0 : n4); m++) {
                const double zz = z[m * 64 + c.lane];
                tl0 = fma(Ri[4 * m + 0], zz, tl0);
                tl1 = fma(Ri[4 * m + 1], zz, tl1);
                tr0 = fma(Ri[4 * m + 2], zz, tr0);
                tr1 = fma(Ri[4 * m + 3], zz, tr1);
            }
        }
        HADI_STAMPB(20);  // reduced system
        // spikes are zero where there is no neighbour (first chunk: V = 0, last chunk: W = 0)
#pragma unroll
        for (int k = 0; k < HADI_LC; k++) {
            y[k] = y[k] - HADI_PB_T(c, k, PB_V0) * tl0 - HADI_PB_T(c, k, PB_V1) * tl1 - HADI_PB_T(c, k, PB_W0) * tr0 -
                   HADI_PB_T(c, k, PB_W1) * tr1;
        }
    }
    HADI_STAMPB(21);  // spike correction
}

// Stores the solved tile `ctile` (with the Ikonen-Toivanen projection for American).  RELOAD: every row's register
// is refilled with the same row of tile `ctile + 1` right behind its store, so one register buffer serves both
// tiles and the loads of the next tile are in flight as soon as the stores have been issued.
template <int AMER, bool RELOAD, class T = double>
HADI_DEV HADI_FORCEINLINE void hadi_pb_store(const HadiPassBCtx &c, int ctile, double (&y)[HADI_LC], int next_tile = -1) {
    static_assert(sizeof(T) == 8 || AMER == 0, "the fp32-state sweep is European only");
    constexpr unsigned ES = (unsigned)sizeof(T);
    HADI_STAMP_DECL(c.stamp_acc_)
    const int coln = (next_tile < 0 ? ctile + 1 : next_tile) * 64 + c.lane;
    const unsigned voffn = (unsigned)(coln < c.rowp ? coln : c.rowp - 1) * ES;
    const int col = ctile * 64 + c.lane;
    const bool valid = col < c.rowp;
    const int colc = valid ? col : c.rowp - 1;
    const size_t base = (size_t)c.ja * c.rowp + colc;
    if constexpr (AMER == 0) {
        const unsigned voff = (unsigned)colc * ES;
        const unsigned rstride = (unsigned)c.rowp * ES;
        if constexpr (RELOAD) {
            const unsigned voffs = valid ? voff : HADI_BUF_DROP;
            // all stores first, then all loads: issued pairwise (store k, load k) the 1024x512 column pass ran 0.142 ms per
            // launch against 0.133 -- the memory pipe turns round between writes and reads 33 times per wavefront and tile
#pragma unroll
            for (int k = 0; k < HADI_LC; k++) hadi_buf_store_t<T>(c.Ub, voffs, hadi_pb_row(c, k) * rstride, y[k]);
#pragma unroll
            for (int k = 0; k < HADI_LC; k++) y[k] = hadi_buf_load_t<T>(c.Yb, voffn, hadi_pb_row(c, k) * rstride);
        } else if (valid) {
#pragma unroll
            for (int k = 0; k < HADI_LC; k++) hadi_buf_store_t<T>(c.Ub, voff, hadi_pb_row(c, k) * rstride, y[k]);
        }
    } else {
        const unsigned row0 = (unsigned)c.ja * (unsigned)c.rowp * 8u, rstride = (unsigned)c.rowp * 8u;
        // Ikonen-Toivanen projection, device_solver.hpp:358-372.  Plain pointer accesses on purpose: with raw buffer
        // operations hipcc cannot tell that the lambda_bar load of row k+1 does not alias the store of row k, keeps them in
        // program order and exposes one memory latency per row (measured 0.204 vs 0.179 ms/launch at 256x128 x512).
        double *__restrict__ dst = c.Ui + base;
        double *__restrict__ Lb = c.Li + base;
        const double *__restrict__ P0 = c.P0i + base;
        const double dt = c.dt;
        const bool is_smax = (col == c.pos_m1);
        // A call / put payoff depends on s only (every driver of the reference builds U_0 that way,
        // heston_calibration.cpp:183-192): then one load per column replaces 33 (8 of the 40 B per node of this pass).
        // Identity padding rows then see the payoff instead of 0: their results are never read.
        const bool pay1d = (AMER == 2) || c.pay1d != 0;
        const double pay_col = pay1d ? c.P0i[colc] : 0.0;
        if constexpr (AMER == 2) {
            // P representation in the single-buffer kernel (round 4): the old P of the tile is fetched in TWO batches of raw
            // buffer loads (17 + 16 rows: with the tile's 33 solved values that is what 128 VGPRs hold), each followed by its
            // projections and stores, then the reloads of the next tile.  Row by row -- load P_old, project, store, reload --
            // every load sat behind the store of the row before it (one array, offsets the compiler cannot tell apart):
            // 1024x512 x64 American puts, column pass 0.203 -> see DESIGN.md section 5.
            constexpr int H0 = (HADI_LC + 1) / 2;
            const unsigned voff = (unsigned)colc * 8u, voffs = valid ? voff : HADI_BUF_DROP;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k0 = h ? H0 : 0, k1 = h ? HADI_LC : H0;
                double po[H0];
#pragma unroll
                for (int k = k0; k < k1; k++) po[k - k0] = hadi_buf_load(c.Ub, voff, row0 + (unsigned)k * rstride);
#pragma unroll
                for (int k = k0; k < k1; k++) {
                    double lamo = fmax(0.0, (pay_col - po[k - k0]) * c.inv_dt);
                    if (is_smax) lamo = 0.0;
                    hadi_buf_store(c.Ub, voffs, row0 + (unsigned)k * rstride, y[k] - dt * lamo);
                }
            }
            if constexpr (RELOAD) {
#pragma unroll
                for (int k = 0; k < HADI_LC; k++) y[k] = hadi_buf_load(c.Yb, voffn, row0 + (unsigned)k * rstride);
            }
            HADI_STAMPB(22);  // projection + store issue
            return;
        }
        // (the explicit (U, lambda_bar) pair in the same two batches was measured and dropped: with the runtime choice between the
        // one-dimensional and the general payoff both loops are live, 122 registers spill, 0.28 -> 0.315 ms per launch)
#pragma unroll
        for (int k = 0; k < HADI_LC; k++) {
            const size_t off = (size_t)k * c.rowp;
            const double U_bar = y[k];
            if constexpr (AMER == 2) {
                // P representation (see hadi_row_step): lambda_bar_old = max(0, (U0 - P_old)/dt), P_new = U_bar - dt lambda_bar_old.
                // One load and one store per node, both on the array that otherwise holds U; no lambda_bar array.
                double lamo = fmax(0.0, (pay_col - dst[off]) * c.inv_dt);
                if (is_smax) lamo = 0.0;
                if (valid) dst[off] = U_bar - dt * lamo;
            } else {
                const double lamv = Lb[off];
                double pay = pay_col;
                if (!pay1d) pay = P0[off];
                const double un = fmax(U_bar - dt * lamv, pay);
                double ln = fmax(0.0, lamv + (pay - U_bar) / dt);
                if (is_smax) ln = 0.0;
                if (valid) {
                    dst[off] = un;
                    Lb[off] = ln;
                }
            }
            if constexpr (RELOAD) y[k] = hadi_buf_load(c.Yb, voffn, row0 + (unsigned)k * rstride);
        }
    }
    HADI_STAMPB(22);  // projection + store issue
}

// American, double-buffered kernel: what the projection of tile `ctile` needs besides the solved values -- the old P
// (P representation) or lambda_bar (explicit pair) -- is fetched into a third register set BEFORE the tile is solved,
// so it arrives during the solve.  Loaded inside the store loop (after the solve's barriers, which no load may cross)
// every tile paid one full memory latency with nothing else to do: 0.28 ms per launch against 0.12 for the European
// column pass on 512x256 x256.
template <int AMER>
HADI_DEV HADI_FORCEINLINE void hadi_pb_load_old(const HadiPassBCtx &c, int ctile, double (&po)[HADI_LC]) {
    const int col = ctile * 64 + c.lane;
    const int colc = col < c.rowp ? col : c.rowp - 1;
    const unsigned voff = (unsigned)colc * 8u;
    const unsigned rstride = (unsigned)c.rowp * 8u;
#pragma unroll
    for (int k = 0; k < HADI_LC; k++) po[k] = hadi_buf_load(AMER == 2 ? c.Ub : c.Lb, voff, hadi_pb_row(c, k) * rstride);
}
// Ikonen-Toivanen projection (device_solver.hpp:358-372) of the solved tile with the prefetched old values; payoff that
// depends on s only (one value per column).  Raw buffer stores; lanes past the pitch are dropped by the range check.
template <int AMER>
HADI_DEV HADI_FORCEINLINE void hadi_pb_store_am(const HadiPassBCtx &c, int ctile, const double (&y)[HADI_LC],
                                                const double (&po)[HADI_LC]) {
    HADI_STAMP_DECL(c.stamp_acc_)
    const int col = ctile * 64 + c.lane;
    const bool valid = col < c.rowp;
    const int colc = valid ? col : c.rowp - 1;
    const unsigned voffs = valid ? (unsigned)colc * 8u : HADI_BUF_DROP;
    const unsigned row0 = (unsigned)c.ja * (unsigned)c.rowp * 8u, rstride = (unsigned)c.rowp * 8u;
    const double dt = c.dt;
    const bool is_smax = (col == c.pos_m1);
    const double pay = c.P0i[colc];
    // (the projection writes the payoff, not zeros, to padding rows: they keep their own storage rows here)
#pragma unroll
    for (int k = 0; k < HADI_LC; k++) {
        const double U_bar = y[k];
        if constexpr (AMER == 2) {
            // lambda_bar_old = max(0, (U0 - P_old)/dt), P_new = U_bar - dt lambda_bar_old (see hadi_row_step)
            double lamo = fmax(0.0, (pay - po[k]) * c.inv_dt);
            if (is_smax) lamo = 0.0;
            hadi_buf_store(c.Ub, voffs, row0 + (unsigned)k * rstride, U_bar - dt * lamo);
        } else {
            const double lamv = po[k];
            const double un = fmax(U_bar - dt * lamv, pay);
            double ln = fmax(0.0, lamv + (pay - U_bar) / dt);
            if (is_smax) ln = 0.0;
            hadi_buf_store(c.Ub, voffs, row0 + (unsigned)k * rstride, un);
            hadi_buf_store(c.Lb, voffs, row0 + (unsigned)k * rstride, ln);
        }
    }
    HADI_STAMPB(22);  // projection + store issue
}

template <int AMER, class T = double>
HADI_DEV HADI_FORCEINLINE void hadi_pb_solve_store(const HadiPassBCtx &c, int ctile, int parity, double (&y)[HADI_LC],
                                                   int younger = 0) {
    if (!(c.debug & HADI_DEBUG_COL_NO_SOLVE)) hadi_pb_solve<false, false, HADI_PB_MF != 0>(c, parity, y, younger);
    hadi_pb_store<AMER, false, T>(c, ctile, y);
}

// Column tiles of block `grp` of an instance.  The pitch is 64*B*G + pad, so the last tile is always a SHORT one (8..32
// columns: little traffic, but a full solve).  The full tiles are dealt out btpw per block and the short tile rides with the
// last block, which is the one that may hold fewer full tiles (1024x512: 16 full tiles on 4 blocks = 4,4,4,4+short instead of
// 5,5,5,2 -- the launch takes 4.3 tile times instead of 5).  With btpw = 1 and one block more than full tiles the short tile
// gets a block of its own (the one-tile-per-block geometry of the small-chunk grids).
// Which full tiles: consecutive ones (block g: g btpw .. g btpw + btpw - 1), or INTERLEAVED (tile_il: block g: g, g + G, g + 2 G,
// ... with G blocks holding full tiles).  The blocks of an instance sit on one XCD in consecutive dispatch slots and walk their
// tiles at the same pace, so interleaved they read -- and write -- G ADJACENT 512-byte segments of every v-row at about the
// same time (one 2 KB piece of a DRAM page per row instead of four pieces 2 KB apart).
struct HadiTileSet {
    int first, stride, nfull_mine, cnt, short_tile;  // tile(i) = i < nfull_mine ? first + i stride : short_tile
};
HADI_DEV HADI_FORCEINLINE HadiTileSet hadi_pb_tiles(const HadiSweepArgs &a, int grp) {
    const int nfull = a.L.rowp >> 6;
    HadiTileSet ts;
    ts.short_tile = nfull;
    int t0 = grp * a.btpw;
    if (t0 > nfull) t0 = nfull;
    const int t1 = (grp == a.bgroups - 1) ? a.ctiles : (t0 + a.btpw < nfull ? t0 + a.btpw : nfull);
    const int has_short = (t1 > nfull) ? 1 : 0;
    if (a.tile_il && nfull > 0) {
        const int gf = (nfull + a.btpw - 1) / a.btpw;  // blocks that hold full tiles
        ts.first = grp; ts.stride = gf;
        ts.nfull_mine = grp < gf ? (nfull - grp + gf - 1) / gf : 0;
    } else {
        ts.first = t0; ts.stride = 1;
        ts.nfull_mine = (t1 < nfull ? t1 : nfull) - t0;
        if (ts.nfull_mine < 0) ts.nfull_mine = 0;
    }
    ts.cnt = ts.nfull_mine + has_short;
    return ts;
}
HADI_DEV HADI_FORCEINLINE int hadi_pb_tile(const HadiTileSet &ts, int i) { return i < ts.nfull_mine ? ts.first + i * ts.stride : ts.short_tile; }

// Dynamic LDS: P * (2*4*64 + 16*P) doubles (two interface-exchange buffers, each wavefront's four rows of the reduced
// inverse); the chunk tables live in registers (HADI_PB_T).
// MAXP only sets the launch bound (register budget): 8 -> 512 threads, 16 -> 1024 threads.
template <int MAXP, int AMER, class T = double>
__global__ void __launch_bounds__(64 * MAXP) hadi_pass_b(HadiSweepArgs a, int n) {
    HADI_DYN_SMEM(double, smem);
    HadiPassBCtx c;
    c.lane = threadIdx.x & 63;
    c.wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    c.P = a.L.P;
    // blocks walk the instances in DESCENDING order: the row pass writes Y ascending, so the column pass starts on the
    // part of Y that is still in the memory-side cache (and leaves the low instances of U there for the next row pass).
    // XCD-aware: the blocks of one instance (they read the same chunk tables and reduced-inverse rows) get consecutive
    // logical ids on ONE XCD, so the second and third reader find the tables in that XCD's L2.
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);
    if (logical >= a.n_inst * a.bgroups) return;  // grid padded to a multiple of 8 (whole block: uniform)
    const int binst = logical / a.bgroups, grp = logical - binst * a.bgroups;
    const int inst = a.n_inst - 1 - binst;
    const HadiInstPar ip = a.ipar[inst];  // (requested here, consumed behind the first tile's loads: see hadi_pass_a_strip)
    const int nrows = a.L.nrows_pad;
    c.nrows = a.L.nrows;
    c.rowp = a.L.rowp;
    c.ja = c.wave * HADI_LC;
    c.Yi = a.Y + (size_t)inst * a.L.inst_stride;  // (pointer-based accesses: American, T = double only)
    c.Ui = a.U + (size_t)inst * a.L.inst_stride;
    c.Yb = hadi_make_buf(reinterpret_cast<const T *>(a.Y) + (size_t)inst * a.L.inst_stride, (size_t)a.L.inst_stride * sizeof(T));
    c.Ub = hadi_make_buf(reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride, (size_t)a.L.inst_stride * sizeof(T));
    c.Li = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride : nullptr;
    c.Lb = hadi_make_buf(c.Li, (AMER == 1) ? (size_t)a.L.inst_stride * sizeof(double) : 0);
    c.P0i = AMER ? a.U0 + (size_t)inst * a.L.inst_stride : nullptr;
    c.pay1d = AMER ? (a.pay_mis[inst] == 0) : 0;
    c.american = a.american; c.debug = a.debug;
    c.pos_m1 = a.pos_m1;
    c.tabl = nullptr;
    const HadiTileSet ts = hadi_pb_tiles(a, grp);
    const int cnt = ts.cnt;
    auto tile = [&](int i) { return hadi_pb_tile(ts, i); };

#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[32] = {0};
    c.stamp_acc_ = stamp_store_;
#endif
    double ya[HADI_LC], yb[HADI_LC];
    hadi_pb_load<T>(c, tile(0), ya);
    const bool am_fast = (AMER == 2) || (AMER == 1 && c.pay1d != 0);  // block-uniform
    // the chunk's table (identical for every column) is spread over the lanes' registers once per block
    hadi_pb_load_table(c, a.pb + ((size_t)inst * nrows + c.ja) * HADI_PBW);
    hadi_pb_setup_lds<HADI_PB_MF != 0>(c, smem, a.rinv + (size_t)inst * 16 * c.P * c.P, 2);
    // the instance's parameters are consumed HERE, behind the first tile's loads and the tables' (hadi_pass_a_strip)
    if (n > ip.N) return;  // whole block: uniform (this instance has fewer time steps; the loads above land in dead registers)
    c.inv_dt = 1.0 / ip.dt;
    c.dt = ip.dt;
    __syncthreads();
    if constexpr (AMER != 0) {
        if (am_fast) {
            double po[HADI_LC];
            for (int i = 0; i < cnt; i += 2) {
                hadi_pb_load_old<AMER>(c, tile(i), po);  // first: it is needed before the next tile's values
                if (i + 1 < cnt) hadi_pb_load<T>(c, tile(i + 1), yb);
                hadi_pb_solve<false, false, HADI_PB_MF != 0>(c, 0, ya, 0);
                hadi_pb_store_am<AMER>(c, tile(i), ya, po);
                if (i + 1 < cnt) {
                    hadi_pb_load_old<AMER>(c, tile(i + 1), po);
                    if (i + 2 < cnt) hadi_pb_load<T>(c, tile(i + 2), ya);
                    hadi_pb_solve<false, false, HADI_PB_MF != 0>(c, 1, yb, 0);
                    hadi_pb_store_am<AMER>(c, tile(i + 1), yb, po);
                }
            }
            return;
        }
    }
    // European: THREE register buffers (3 x 33 rows = 198 VGPRs of 256) -- the loads of two tiles are in flight while one
    // is solved and stored.  With two buffers a tile's loads had only the short solve of its predecessor (~4 k cycles)
    // to arrive in, less than the memory latency under load; measured 512x256 x256: 0.117 -> 0.113 ms per launch at the
    // same 3 tiles per block (a block then has all its loads in flight from the start).
    if constexpr (AMER == 0) {
        double yc[HADI_LC];
        if (1 < cnt) hadi_pb_load<T>(c, tile(1), yb);
        for (int i = 0; i < cnt; i += 3) {
            if (i + 2 < cnt) hadi_pb_load<T>(c, tile(i + 2), yc);
            hadi_pb_solve_store<AMER, T>(c, tile(i), i & 1, ya, 0);
            if (i + 1 < cnt) {
                if (i + 3 < cnt) hadi_pb_load<T>(c, tile(i + 3), ya);
                hadi_pb_solve_store<AMER, T>(c, tile(i + 1), (i + 1) & 1, yb, 0);
            }
            if (i + 2 < cnt) {
                if (i + 4 < cnt) hadi_pb_load<T>(c, tile(i + 4), yb);
                hadi_pb_solve_store<AMER, T>(c, tile(i + 2), (i + 2) & 1, yc, 0);
            }
        }
        return;
    }
    for (int i = 0; i < cnt; i += 2) {  // American with a payoff that depends on v: two buffers, loads inside the store loop
        // `younger` = vector-memory operations issued after the loads of the tile being solved (diagnostic build only)
        if (i + 1 < cnt) hadi_pb_load<T>(c, tile(i + 1), yb);
        hadi_pb_solve_store<AMER, T>(c, tile(i), 0, ya, (i + 1 < cnt ? HADI_LC : 0) + (i > 0 ? HADI_LC : 0));
        if (i + 2 < cnt) hadi_pb_load<T>(c, tile(i + 2), ya);
        if (i + 1 < cnt) hadi_pb_solve_store<AMER, T>(c, tile(i + 1), 1, yb, (i + 2 < cnt ? HADI_LC : 0) + HADI_LC);
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (HADI_STAMPS == 3 && c.lane == 0)
        for (int k = 16; k < 24; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
}

// Single-buffer variant for more than 8 chunks (m2 > 263): a 1024-thread block has 128 VGPRs per lane, too few for two
// 33-row register buffers (the double-buffered code spills 650 B per lane there).  The next tile is loaded into the
// registers of the current one row by row, right behind the stores.  Measured on MI355X: 1024x512 grid 0.250 ms per
// launch against 0.382; at 512x256 (P = 8) the double-buffered kernel above wins, 0.144 against 0.206.
template <int MAXP, int AMER, class T = double>
__global__ void __launch_bounds__(64 * MAXP, 4) hadi_pass_b1(HadiSweepArgs a, int n) {
    HADI_DYN_SMEM(double, smem);
    HadiPassBCtx c;
    c.lane = threadIdx.x & 63;
    c.wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    c.P = a.L.P;
    // blocks walk the instances in DESCENDING order: the row pass writes Y ascending, so the column pass starts on the
    // part of Y that is still in the memory-side cache (and leaves the low instances of U there for the next row pass).
    // XCD-aware: the blocks of one instance (they read the same chunk tables and reduced-inverse rows) get consecutive
    // logical ids on ONE XCD, so the second and third reader find the tables in that XCD's L2.
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);
    if (logical >= a.n_inst * a.bgroups) return;  // grid padded to a multiple of 8 (whole block: uniform)
    const int binst = logical / a.bgroups, grp = logical - binst * a.bgroups;
    const int inst = a.n_inst - 1 - binst;
    const HadiInstPar ip = a.ipar[inst];  // (requested here, consumed behind the first tile's loads: see hadi_pass_a_strip)
    const int nrows = a.L.nrows_pad;
    c.nrows = a.L.nrows;
    c.rowp = a.L.rowp;
    c.ja = c.wave * HADI_LC;
    c.Yi = a.Y + (size_t)inst * a.L.inst_stride;  // (pointer-based accesses: American, T = double only)
    c.Ui = a.U + (size_t)inst * a.L.inst_stride;
    c.Yb = hadi_make_buf(reinterpret_cast<const T *>(a.Y) + (size_t)inst * a.L.inst_stride, (size_t)a.L.inst_stride * sizeof(T));
    c.Ub = hadi_make_buf(reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride, (size_t)a.L.inst_stride * sizeof(T));
    c.Li = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride : nullptr;
    c.Lb = hadi_make_buf(c.Li, (AMER == 1) ? (size_t)a.L.inst_stride * sizeof(double) : 0);
    c.P0i = AMER ? a.U0 + (size_t)inst * a.L.inst_stride : nullptr;
    c.pay1d = AMER ? (a.pay_mis[inst] == 0) : 0;
    c.american = a.american; c.debug = a.debug;
    c.pos_m1 = a.pos_m1;
    c.tabl = nullptr;
    const HadiTileSet ts = hadi_pb_tiles(a, grp);
    const int cnt = ts.cnt;
    auto tile = [&](int i) { return hadi_pb_tile(ts, i); };

#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[32] = {0};
    c.stamp_acc_ = stamp_store_;
#endif
    double y[HADI_LC];
    hadi_pb_load<T>(c, tile(0), y);
    // the chunk's table (identical for every column) is spread over the lanes' registers once per block
    hadi_pb_load_table(c, a.pb + ((size_t)inst * nrows + c.ja) * HADI_PBW);
    hadi_pb_setup_lds<HADI_PB_MF != 0>(c, smem, a.rinv + (size_t)inst * 16 * c.P * c.P, 2);
    if (n > ip.N) return;  // whole block: uniform (consumed behind the first tile's loads and the tables')
    c.inv_dt = 1.0 / ip.dt;
    c.dt = ip.dt;
    __syncthreads();
    // (fp32 state: holding the NEXT tile in 33 float registers so that its loads fly during the solve was tried -- 66 + 33 +
    // 10 table registers leave too few of the 128 for the reduced-system loop, the kernel spills 18 registers and the
    // scratch reloads drain the prefetch: 0.088 -> 0.099 ms per launch at 1024x512 x64.)
    for (int i = 0; i < cnt; i++) {
        if (!(a.debug & HADI_DEBUG_COL_NO_SOLVE)) hadi_pb_solve<false, false, HADI_PB_MF != 0>(c, i & 1, y, 0);
        if (i + 1 < cnt) hadi_pb_store<AMER, true, T>(c, tile(i), y, tile(i + 1));
        else hadi_pb_store<AMER, false, T>(c, tile(i), y);
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (HADI_STAMPS == 3 && c.lane == 0)
        for (int k = 16; k < 24; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
}

// ------------------------------------------------------------------------------------------------
// hadi_pass_b2: the single-buffer column pass with part of the NEXT tile prefetched into LDS (9 .. 16 chunks, European).
// hadi_pass_b1 cannot hold a second tile -- 16 wavefronts per CU leave 128 VGPRs per lane, and a 16-chunk tile is 263 KB
// against the 512 KB of a CU's whole register file -- so it stores a tile, loads the next one behind the stores and waits:
// every tile pays the turn-round of the memory pipe and a full load latency with nothing in flight during the solve
// (PMC: VALU busy 0.08, waiting 0.37 of the wave cycles).  LDS-DMA needs no registers: here every wavefront fetches the first
// NPF rows of its chunk of tile t+1 into a private LDS area BEFORE it solves tile t, so they fly during the solve and the
// stores; only the other 33 - NPF rows are loaded into the registers behind the stores.  The LDS comes from the second
// exchange buffer (one buffer + a second barrier per tile: hadi_pb_solve<.., ONEBUF>): 16 chunks: 32 KB exchange + 32 KB
// reduced-inverse rows + 16 x NPF x 64 elements.  One dwordx4 DMA instruction moves 64 x 16 B = RPI rows of this tile (a row
// is 64 columns = 512 B as doubles, 256 B as floats): RPI = 2 resp. 4 rows, lanes grouped by row.
// Completion: the wavefront's explicit s_waitcnt vmcnt(0) behind the register loads (they are younger than the DMA and the
// stores; everything has to be there before the solve anyway) -- no counted waits, nothing depends on the retirement order
// of different kinds of vector-memory operations.
template <class T, int NPF>
HADI_DEV HADI_FORCEINLINE void hadi_pb_dma(const HadiPassBCtx &c, const T *__restrict__ Yt, int ctile, T *pf) {
    constexpr int ES = (int)sizeof(T), EPV = 16 / ES, LPR = 64 / EPV, RPI = 64 / LPR;
    static_assert(NPF % RPI == 0 && NPF <= HADI_LC, "whole DMA instructions");
    const int sub = c.lane / LPR, l = c.lane - sub * LPR;
    int col = ctile * 64 + EPV * l;
    if (col + EPV > c.rowp) col = c.rowp - EPV;  // lanes past the pitch (short tile) fetch a valid address; their columns are never stored
#if !defined(HADI_EMU)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)pf);
#endif
#pragma unroll
    for (int q = 0; q < NPF / RPI; q++) {
        const unsigned row = hadi_pb_row(c, q * RPI + sub);  // (per lane: the lanes of one instruction cover RPI rows)
#if defined(HADI_EMU)
        for (int e = 0; e < EPV; e++) pf[(q * RPI + sub) * 64 + EPV * l + e] = Yt[(size_t)row * c.rowp + col + e];
#else
        const unsigned voff = (row * (unsigned)c.rowp + (unsigned)col) * (unsigned)ES;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(Yt), "s"(lds0 + 1024u * q)
                     : "memory");
#endif
    }
}
// rows [K0, HADI_LC) of tile `ctile` into the registers (hadi_pb_load for a part of the chunk)
template <class T, int K0>
HADI_DEV HADI_FORCEINLINE void hadi_pb_load_from(const HadiPassBCtx &c, int ctile, double (&y)[HADI_LC]) {
    constexpr unsigned ES = (unsigned)sizeof(T);
    const int col = ctile * 64 + c.lane;
    const int colc = col < c.rowp ? col : c.rowp - 1;
    const unsigned voff = (unsigned)colc * ES;
    const unsigned rstride = (unsigned)c.rowp * ES;
#pragma unroll
    for (int k = K0; k < HADI_LC; k++) y[k] = hadi_buf_load_t<T>(c.Yb, voff, hadi_pb_row(c, k) * rstride);
}

template <int MAXP, class T, int NPF>
__global__ void __launch_bounds__(64 * MAXP, 4) hadi_pass_b2(HadiSweepArgs a, int n) {
    HADI_DYN_SMEM(double, smem);
    HadiPassBCtx c;
    c.lane = threadIdx.x & 63;
    c.wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    c.P = a.L.P;
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);  // (block order and XCD placement as in hadi_pass_b)
    if (logical >= a.n_inst * a.bgroups) return;
    const int binst = logical / a.bgroups, grp = logical - binst * a.bgroups;
    const int inst = a.n_inst - 1 - binst;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;
    const int nrows = a.L.nrows_pad;
    c.nrows = a.L.nrows;
    c.rowp = a.L.rowp;
    c.ja = c.wave * HADI_LC;
    c.Yi = a.Y + (size_t)inst * a.L.inst_stride;
    c.Ui = a.U + (size_t)inst * a.L.inst_stride;
    const T *__restrict__ Yt = reinterpret_cast<const T *>(a.Y) + (size_t)inst * a.L.inst_stride;
    c.Yb = hadi_make_buf(Yt, (size_t)a.L.inst_stride * sizeof(T));
    c.Ub = hadi_make_buf(reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride, (size_t)a.L.inst_stride * sizeof(T));
    c.Li = nullptr; c.Lb = hadi_make_buf(nullptr, 0); c.P0i = nullptr; c.pay1d = 0;
    c.inv_dt = 1.0 / ip.dt;
    c.american = 0; c.debug = a.debug;
    c.pos_m1 = a.pos_m1;
    c.dt = ip.dt;
    c.tabl = nullptr;
    const HadiTileSet ts = hadi_pb_tiles(a, grp);
    const int cnt = ts.cnt;
    auto tile = [&](int i) { return hadi_pb_tile(ts, i); };
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[32] = {0};
    c.stamp_acc_ = stamp_store_;
#endif
    double y[HADI_LC];
    hadi_pb_load<T>(c, tile(0), y);
    hadi_pb_load_table(c, a.pb + ((size_t)inst * nrows + c.ja) * HADI_PBW);
    // ONE exchange buffer either way; this wavefront's prefetch area behind the reduced system's LDS
    T *const pf = reinterpret_cast<T *>(hadi_pb_setup_lds<HADI_PB_MF != 0>(c, smem, a.rinv + (size_t)inst * 16 * c.P * c.P, 1)) + (size_t)c.wave * NPF * 64;
    __syncthreads();
    for (int i = 0; i < cnt; i++) {
        const bool more = i + 1 < cnt;  // (block-uniform)
        const int t = tile(i), tn = tile(more ? i + 1 : i);
        if (more) hadi_pb_dma<T, NPF>(c, Yt, tn, pf);  // flies during the solve and the stores of tile t
        if (!(a.debug & HADI_DEBUG_COL_NO_SOLVE)) hadi_pb_solve<false, true, HADI_PB_MF != 0>(c, 0, y, 0);
        if (more) {
            hadi_pb_store<0, false, T>(c, t, y);
            hadi_pb_load_from<T, NPF>(c, tn, y);
#if !defined(HADI_EMU)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA (older than everything above) has landed
#endif
            hadi_wave_rendezvous();
#pragma unroll
            for (int k = 0; k < NPF; k++) y[k] = (double)pf[k * 64 + c.lane];
#if !defined(HADI_EMU)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and has been read: the next DMA may overwrite the area
#endif
            hadi_wave_rendezvous();
        } else {
            hadi_pb_store<0, false, T>(c, t, y);
        }
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (HADI_STAMPS == 3 && c.lane == 0)
        for (int k = 16; k < 24; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
}

