// emu_driver.cpp -- TEST-ONLY.  Runs csrc/hadi_kernels.h (the product's kernel source, unmodified)
// under the host-thread wave emulator so tests can compare its logic with the oracle without a GPU.
// Never shipped, never linked into libhadi; see wave_emu.h.
#define HADI_EMU 1
#include "hadi_kernels.h"
#include "hadi_plan.h"

#include <string>
#include <vector>

namespace emu {
thread_local emu_dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockState *t_block;
thread_local WaveState *t_wave;
thread_local int t_lane;
}  // namespace emu

// kernel-selection overrides, the emulator's stand-in for hadi_set_tuning
static HadiTuning g_tune;
static int g_err = 0, g_debug = 0;  // the handle's device error word and the "debug_fault" test hook
static int g_tile_il = 0;           // "tile_interleave": the column pass's blocks take their full tiles interleaved (opt-in, as in the library)
static int g_cs_strips = 1;         // "cs_strips": Craig-Sneyd row passes on strips where the plan chose strips (default, as in the library)
static int g_team_blocks = 1;       // "team_blocks": blocks per team of the instance-resident launch (> 1: the grid's blocks run concurrently)
static int g_col_prefetch = 0;      // "col_prefetch": hadi_pass_b2 for European sweeps of 9 .. 16 chunks (opt-in, as in the library)
extern "C" int emu_take_error() { const int e = g_err; g_err = 0; return e; }
extern "C" int emu_set_tuning(const char *key, int value) {
    const std::string k(key);
    if (k == "debug_fault") g_debug = value;
    else if (k == "strip") g_tune.strip = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "row_tile") g_tune.row_tile = value > 0 ? value : 0;
    else if (k == "strip_blocks") g_tune.strip_blocks = value > 0 ? value : 0;
    else if (k == "pair_strips") g_tune.pair_strips = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "col_groups") g_tune.col_groups = value > 0 ? value : 0;
    else if (k == "col_prefetch") g_col_prefetch = value ? 1 : 0;
    else if (k == "cs_strips") g_cs_strips = value ? 1 : 0;
    else if (k == "team_blocks") g_team_blocks = value > 0 ? value : 1;
    else if (k == "tile_interleave") g_tile_il = value ? 1 : 0;
    else if (k == "reset") { g_tune = HadiTuning(); g_debug = 0; g_col_prefetch = 0; g_tile_il = 0; g_cs_strips = 1; g_team_blocks = 1; }
    else return 1;
    return 0;
}

template <int B, int G, int NG, int PD>
static void run_pass_a(const HadiPlan &pl, const HadiSweepArgs &a, int n, int mode) {
    const unsigned nt = 64 * pl.W * G * NG;
    if (mode == 1) emu::launch(pl.grid_a, nt, [&]() { hadi_pass_a<B, G, 4, NG, PD, false, 1>(a, n); }, pl.smem_a);
    else if (mode == 2) emu::launch(pl.grid_a, nt, [&]() { hadi_pass_a<B, G, 4, NG, PD, false, 2>(a, n); }, pl.smem_a);
    else if (a.american) emu::launch(pl.grid_a, nt, [&]() { hadi_pass_a<B, G, 4, NG, PD, true>(a, n); }, pl.smem_a);
    else emu::launch(pl.grid_a, nt, [&]() { hadi_pass_a<B, G, 4, NG, PD, false>(a, n); }, pl.smem_a);
}

static int run_row_pass(const HadiPlan &pl, const HadiSweepArgs &a, int n, int mode) {
    if (pl.row_seq) {  // same choice as hadi_api.hip
        if (mode) return 2;
        const unsigned g = a.n_inst * ((pl.L.nrows + 63) / 64);
        if (a.american) emu::launch(g, 64, [&]() { hadi_pass_a_seq<1>(a, n); });
        else emu::launch(g, 64, [&]() { hadi_pass_a_seq<0>(a, n); });
        return 0;
    }
    if (pl.use_pairs && pl.use_strip && mode == 0) {  // two strips per wavefront (4 nodes per lane)
        if (a.american) emu::launch(pl.grid_as, 64 * HADI_PAIR_WAVES, [&]() { hadi_pass_a_pairs<1>(a, n); }, pl.smem_pairs_eu);
        else emu::launch(pl.grid_as, 64 * HADI_PAIR_WAVES, [&]() { hadi_pass_a_pairs<0>(a, n); }, pl.smem_pairs_eu);
        return 0;
    }
    if (pl.use_strip && mode != 0 && !pl.use_pairs && g_cs_strips) {  // Craig-Sneyd on strips: same choice as hadi_api.hip
        const unsigned nt = 64 * HADI_STRIP_WAVES(pl.L.B);
        if (pl.L.G == 2) {
            if (mode == 1) emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<8, 0, double, 2, 1>(a, n); }, pl.smem_as);
            else emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<8, 0, double, 2, 2>(a, n); }, pl.smem_as);
            return 0;
        }
        switch (pl.L.B * 4 + mode) {
            case 33: emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<8, 0, double, 1, 1>(a, n); }, pl.smem_as); break;
            case 34: emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<8, 0, double, 1, 2>(a, n); }, pl.smem_as); break;
            case 17: emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<4, 0, double, 1, 1>(a, n); }, pl.smem_as); break;
            case 18: emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<4, 0, double, 1, 2>(a, n); }, pl.smem_as); break;
            case 9: emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<2, 0, double, 1, 1>(a, n); }, pl.smem_as); break;
            default: emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<2, 0, double, 1, 2>(a, n); }, pl.smem_as); break;
        }
        return 0;
    }
    if (pl.use_strip && mode == 0 && pl.L.G == 2) {  // paired strips
        if (a.american) emu::launch(pl.grid_as, 512, [&]() { hadi_pass_a_strip<8, 1, double, 2>(a, n); }, pl.smem_as);
        else emu::launch(pl.grid_as, 512, [&]() { hadi_pass_a_strip<8, false, double, 2>(a, n); }, pl.smem_as);
        return 0;
    }
    if (pl.use_strip && mode == 0 && pl.L.G == 1) {  // same choice as hadi_api.hip
        switch (pl.L.B * 2 + (a.american ? 1 : 0)) {
            case 16: emu::launch(pl.grid_as, 512, [&]() { hadi_pass_a_strip<8, false>(a, n); }, pl.smem_as); break;
            case 17: emu::launch(pl.grid_as, 512, [&]() { hadi_pass_a_strip<8, true>(a, n); }, pl.smem_as); break;
            case 8: emu::launch(pl.grid_as, 64 * HADI_STRIP_WAVES(4), [&]() { hadi_pass_a_strip<4, false>(a, n); }, pl.smem_as); break;
            case 9: emu::launch(pl.grid_as, 64 * HADI_STRIP_WAVES(4), [&]() { hadi_pass_a_strip<4, true>(a, n); }, pl.smem_as); break;
            case 4: emu::launch(pl.grid_as, 64 * HADI_STRIP_WAVES(2), [&]() { hadi_pass_a_strip<2, false>(a, n); }, pl.smem_as); break;
            default: emu::launch(pl.grid_as, 64 * HADI_STRIP_WAVES(2), [&]() { hadi_pass_a_strip<2, true>(a, n); }, pl.smem_as); break;
        }
        return 0;
    }
    switch (pl.L.B * 10 + pl.L.G) {
        case 11: run_pass_a<1, 1, 1, 2>(pl, a, n, mode); break;
        case 21: run_pass_a<2, 1, 1, 2>(pl, a, n, mode); break;
        case 41: run_pass_a<4, 1, 1, 2>(pl, a, n, mode); break;
        case 81: run_pass_a<8, 1, 1, 1>(pl, a, n, mode); break;
        case 82: run_pass_a<8, 2, 1, 1>(pl, a, n, mode); break;
        default: return 2;
    }
    return 0;
}

// fp32-state sweep (scheme == 2 in emu_solve): same choices as hadi_api.hip
template <int B, int G, int NG, int PD>
static void run_pass_a_f32(const HadiPlan &pl, const HadiSweepArgs &a, int n) {
    const size_t ring_elems = (size_t)NG * ((PD + 1) * pl.W + 4) * pl.L.rowp;
    emu::launch(pl.grid_a, 64 * pl.W * G * NG, [&]() { hadi_pass_a<B, G, 4, NG, PD, false, 0, float>(a, n); },
                pl.smem_a - ring_elems * (sizeof(double) - sizeof(float)));
}
static int run_sweep_f32(const HadiPlan &pl, const HadiSweepArgs &a, int n) {
    if (pl.use_strip && pl.L.B == 8 && pl.L.G == 2) {
        emu::launch(pl.grid_as, 512, [&]() { hadi_pass_a_strip<8, false, float, 2>(a, n); }, pl.smem_as);
    } else if (pl.use_strip && pl.L.B == 8) {
        emu::launch(pl.grid_as, 512, [&]() { hadi_pass_a_strip<8, false, float>(a, n); },
                    (size_t)8 * 4 * pl.L.rowp * sizeof(float) + (size_t)4 * 64 * pl.L.B * sizeof(double));
    } else switch (pl.L.B * 10 + pl.L.G) {
        case 11: run_pass_a_f32<1, 1, 1, 2>(pl, a, n); break;
        case 21: run_pass_a_f32<2, 1, 1, 2>(pl, a, n); break;
        case 41: run_pass_a_f32<4, 1, 1, 2>(pl, a, n); break;
        case 81: run_pass_a_f32<8, 1, 1, 1>(pl, a, n); break;
        case 82: run_pass_a_f32<8, 2, 1, 1>(pl, a, n); break;
        default: return 2;
    }
    if (pl.L.P <= 8) emu::launch(pl.grid_b, pl.block_b, [&]() { hadi_pass_b<8, false, float>(a, n); }, pl.smem_b);
    else if (g_col_prefetch) emu::launch(pl.grid_b, pl.block_b, [&]() { hadi_pass_b2<16, float, HADI_B2_NPF(4)>(a, n); }, pl.smem_b2);
    else emu::launch(pl.grid_b, pl.block_b, [&]() { hadi_pass_b1<16, false, float>(a, n); }, pl.smem_b);
    return 0;
}

// American, P representation (scheme == 3 in emu_solve): same kernels as hadi_api.hip picks
template <int B, int G, int NG, int PD>
static void run_pass_a_amp(const HadiPlan &pl, const HadiSweepArgs &a, int n) {
    emu::launch(pl.grid_a, 64 * pl.W * G * NG, [&]() { hadi_pass_a<B, G, 4, NG, PD, 2>(a, n); },
                pl.smem_a + (size_t)pl.L.rowp * sizeof(double));
}
static int run_sweep_amp(const HadiPlan &pl, const HadiSweepArgs &a, int n) {
    if (pl.use_pairs && pl.use_strip) {
        emu::launch(pl.grid_as, 64 * HADI_PAIR_WAVES, [&]() { hadi_pass_a_pairs<2>(a, n); }, pl.smem_pairs_amp);
    } else if (pl.use_strip && pl.L.G == 2) {  // paired strips
        emu::launch(pl.grid_as, 512, [&]() { hadi_pass_a_strip<8, 2, double, 2>(a, n); }, pl.smem_as + (size_t)pl.L.rowp * sizeof(double));
    } else if (pl.use_strip && pl.L.G == 1) {  // same choice as hadi_api.hip
        const unsigned nt = 64 * HADI_STRIP_WAVES(pl.L.B);
        const size_t sm = pl.smem_as + (size_t)pl.L.rowp * sizeof(double);
        switch (pl.L.B) {
            case 8: emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<8, 2>(a, n); }, sm); break;
            case 4: emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<4, 2>(a, n); }, sm); break;
            default: emu::launch(pl.grid_as, nt, [&]() { hadi_pass_a_strip<2, 2>(a, n); }, sm); break;
        }
    } else switch (pl.L.B * 10 + pl.L.G) {
        case 11: run_pass_a_amp<1, 1, 1, 2>(pl, a, n); break;
        case 21: run_pass_a_amp<2, 1, 1, 2>(pl, a, n); break;
        case 41: run_pass_a_amp<4, 1, 1, 2>(pl, a, n); break;
        case 81: run_pass_a_amp<8, 1, 1, 1>(pl, a, n); break;
        case 82: run_pass_a_amp<8, 2, 1, 1>(pl, a, n); break;
        default: return 2;
    }
    if (pl.L.P <= 8) emu::launch(pl.grid_b, pl.block_b, [&]() { hadi_pass_b<8, 2>(a, n); }, pl.smem_b);
    else emu::launch(pl.grid_b, pl.block_b, [&]() { hadi_pass_b1<16, 2>(a, n); }, pl.smem_b);
    return 0;
}

static void run_col_pass(const HadiPlan &pl, const HadiSweepArgs &a, int n) {
    if (pl.col_seq) {  // same choice as hadi_api.hip
        const unsigned g = a.n_inst * pl.ctiles;
        if (a.american) emu::launch(g, 64, [&]() { hadi_pass_b_seq<1>(a, n); });
        else emu::launch(g, 64, [&]() { hadi_pass_b_seq<0>(a, n); });
        return;
    }
    if (pl.L.P <= 8) {  // same choice as hadi_api.hip
        if (a.american) emu::launch(pl.grid_b, pl.block_b, [&]() { hadi_pass_b<8, true>(a, n); }, pl.smem_b);
        else emu::launch(pl.grid_b, pl.block_b, [&]() { hadi_pass_b<8, false>(a, n); }, pl.smem_b);
    } else {
        if (a.american) emu::launch(pl.grid_b, pl.block_b, [&]() { hadi_pass_b1<16, true>(a, n); }, pl.smem_b);
        else if (g_col_prefetch) emu::launch(pl.grid_b, pl.block_b, [&]() { hadi_pass_b2<16, double, HADI_B2_NPF(8)>(a, n); }, pl.smem_b2);
        else emu::launch(pl.grid_b, pl.block_b, [&]() { hadi_pass_b1<16, false>(a, n); }, pl.smem_b);
    }
}

extern "C" int emu_plan(int m1, int m2, int n_inst, int target_waves, int *out /*B,rowp,P,R,ntiles,ctiles*/) {
    HadiPlan pl;
    if (hadi_make_plan(m1, m2, n_inst, target_waves, &pl, g_tune)) return 1;
    out[0] = pl.L.B; out[1] = pl.L.rowp; out[2] = pl.L.P; out[3] = pl.R; out[4] = pl.ntiles; out[5] = pl.L.G;
    return 0;
}

// every launch-geometry field of the plan, for the host-logic invariants test
extern "C" int emu_plan_full(int m1, int m2, int n_inst, int target_waves, long long *o /*[26]*/) {
    HadiPlan pl;
    if (hadi_make_plan(m1, m2, n_inst, target_waves, &pl, g_tune)) return 1;
    const HadiLayout &L = pl.L;
    o[0] = L.B; o[1] = L.G; o[2] = L.rowp; o[3] = L.P; o[4] = L.nrows; o[5] = L.nrows_pad; o[6] = L.inst_stride;
    o[7] = pl.W; o[8] = pl.NG; o[9] = pl.PD; o[10] = pl.R; o[11] = pl.ntiles; o[12] = pl.grid_a; o[13] = (long long)pl.smem_a;
    o[14] = pl.use_strip; o[15] = pl.RS; o[16] = pl.sblocks; o[17] = pl.grid_as; o[18] = (long long)pl.smem_as;
    o[19] = pl.ctiles; o[20] = pl.btpw; o[21] = pl.bgroups; o[22] = pl.grid_b; o[23] = (long long)pl.smem_b;
    o[24] = pl.use_pairs; o[25] = (long long)pl.smem_pairs_amp;
    return 0;
}

// the column tiles block `grp` of an instance walks, in order (hadi_pb_tiles); returns their number
extern "C" int emu_col_tiles(int m1, int m2, int n_inst, int target_waves, int interleave, int grp, int *out /*[ctiles]*/) {
    HadiPlan pl;
    if (hadi_make_plan(m1, m2, n_inst, target_waves, &pl, g_tune)) return -1;
    HadiSweepArgs a{};
    a.L = pl.L; a.ctiles = pl.ctiles; a.btpw = pl.btpw; a.bgroups = pl.bgroups; a.tile_il = interleave;
    if (grp >= pl.bgroups) return -2;
    const HadiTileSet ts = hadi_pb_tiles(a, grp);
    for (int i = 0; i < ts.cnt; i++) out[i] = hadi_pb_tile(ts, i);
    return ts.cnt;
}

// the fraction of its CU-rounds the plan's row pass leaves idle (decides one or two streams: hadi_plan_row_idle), x 1e6
extern "C" long long emu_plan_row_idle_ppm(int m1, int m2, int n_inst, int cus) {
    HadiPlan pl;
    if (hadi_make_plan(m1, m2, n_inst, 8 * cus, &pl, g_tune)) return -1;
    return (long long)(hadi_plan_row_idle(pl, n_inst, cus) * 1e6 + 0.5);
}

// variant bit0 = american, bit1 = dividends.  Arrays natural layout [n][...].
extern "C" int emu_solve(int n_inst, int m1, int m2, int N, double dt, double theta, double r_d, double r_f,
                         const double *par /*[n][4] rho sigma kappa eta*/, int variant, const double *vec_s,
                         const double *vec_v, const double *delta_s, const double *delta_v, double *U,
                         const double *U0, double *lam_out, int target_waves, int ndiv, const double *ddates,
                         const double *damounts, const double *dpcts, int setup_threads, int use_small, int scheme,
                         const double *put_strikes /* NULL = call boundary data */) {
    HadiPlan pl;
    if (hadi_make_plan(m1, m2, n_inst, target_waves, &pl, g_tune, scheme == 2 ? 4 : 8)) return 1;
    const HadiLayout &L = pl.L;
    const int american = variant & 1, dividend = (variant >> 1) & 1;
    const size_t st = (size_t)L.inst_stride * n_inst;
    const bool cs = scheme == 1, f32 = scheme == 2, amp = scheme == 3;
    if (amp && !american) return 3;
    if (f32 && (american || dividend)) return 3;
    std::vector<double> dV(cs ? st : 0), dR1(cs ? st : 0), dC2(cs ? st : 0);
    std::vector<double> dU(st), dY(st, 0.0), dLAM(american ? st : 0), dU0(american ? st : 0), dUT(dividend ? st : 0);
    std::vector<double> scoef(pl.n_scoef * n_inst), b2row(pl.n_b2row * n_inst), rowc(pl.n_rowc * n_inst),
        a2i(pl.n_a2i * n_inst), pb(pl.n_pb * n_inst), rinv(pl.n_rinv * n_inst), rwork(pl.n_rwork * n_inst);
    std::vector<HadiInstPar> ipar(n_inst);
    std::vector<double> par8((size_t)n_inst * 8);
    for (int k = 0; k < n_inst; k++) {
        for (int z = 0; z < 4; z++) par8[(size_t)k * 8 + z] = par[(size_t)k * 4 + z];
        par8[(size_t)k * 8 + 4] = dt;
        par8[(size_t)k * 8 + 5] = (double)N;
        par8[(size_t)k * 8 + 6] = put_strikes ? put_strikes[k] : 0.0;
        par8[(size_t)k * 8 + 7] = put_strikes ? 1.0 : 0.0;
    }
    HadiSetupArgs s;
    s.L = L; s.n_inst = n_inst;
    s.vec_s = vec_s; s.vec_v = vec_v; s.delta_s = delta_s; s.delta_v = delta_v;
    s.par = par8.data(); s.r_d = r_d; s.r_f = r_f; s.theta = theta;
    s.scoef = scoef.data(); s.b2row = b2row.data(); s.rowc = rowc.data(); s.a2i = a2i.data();
    s.pb = pb.data(); s.rinv = rinv.data(); s.rwork = rwork.data(); s.ipar = ipar.data();
    emu::launch(n_inst, setup_threads, [&]() { hadi_setup_kernel(s); });

    emu::launch(8, 64, [&]() { hadi_pack_kernel(L, n_inst, n_inst, U, dU.data()); });
    if (american) {
        emu::launch(8, 64, [&]() { hadi_pack_kernel(L, n_inst, n_inst, U0 ? U0 : U, dU0.data()); });
        emu::launch(8, 64, [&]() { hadi_fill_kernel(dLAM.data(), st, 0.0); });
    }
    HadiSweepArgs a;
    a.U = dU.data(); a.Y = dY.data(); a.LAM = american ? dLAM.data() : nullptr; a.U0 = american ? dU0.data() : nullptr;
    a.scoef = scoef.data(); a.b2row = b2row.data(); a.rowc = rowc.data(); a.pb = pb.data(); a.rinv = rinv.data();
    a.ipar = ipar.data(); a.L = L; a.n_inst = n_inst; a.R = pl.R; a.ntiles = pl.ntiles; a.ctiles = pl.ctiles; a.btpw = pl.btpw; a.bgroups = pl.bgroups; a.tile_il = g_tile_il;
    a.american = american; a.pos_m1 = pl.pos_m1; a.RS = pl.RS; a.sblocks = pl.sblocks;
    a.err = &g_err; a.debug = g_debug;
    std::vector<int> pay_mis(n_inst, 0);
    a.pay_mis = american ? pay_mis.data() : nullptr;
    if (american) emu::launch(8, 64, [&]() { hadi_payoff_shape_kernel(L, n_inst, dU0.data(), pay_mis.data()); });
    std::vector<double> dW(pl.row_seq ? st : 0);
    a.R1 = cs ? dR1.data() : pl.row_seq ? dW.data() : nullptr; a.C2 = cs ? dC2.data() : nullptr;
    const bool pair_tab = pl.L.G == 2 && !cs && !pl.row_seq && pl.use_strip;  // as hadi_api.hip: the pairs' coupling column, built once
    std::vector<double> dRS(pair_tab ? (size_t)n_inst * pl.L.nrows * 128 : 0, std::nan(""));
    a.rs_tab = pair_tab ? dRS.data() : nullptr;
    HadiSweepArgs av = a;
    if (cs) av.U = dV.data();

    std::vector<int> flags(N, -1);
    if (dividend) hadi_dividend_steps(N, dt, ndiv, ddates, flags.data(), N);
    if (use_small == 4) {  // instance-resident launch (hadi_team_kernel) with teams of ONE block: the emulator runs the blocks of a
                           // grid one after the other, so the team barrier is trivially met; indexing and arithmetic are real
        if (american || cs || f32 || n_inst > 8 || L.G != 1 || (L.B != 8 && L.B != 4) || L.P > 8) return 3;
        std::vector<int> team(512, 0);
        HadiTeamArgs ta;
        // "team_blocks" > 1: teams of several blocks, all blocks of the grid running at once (the team barrier, the formation
        // counters and the row / column-tile split over the blocks are then real); 1: block after block, a team per block
        const int nb = g_team_blocks;
        ta.form = team.data(); ta.bar = team.data() + 64; ta.nb = nb; ta.N = N; ta.stamps = nullptr;
        ta.div_flag = dividend ? flags.data() : nullptr; ta.flag_stride = 0; ta.div_amounts = damounts; ta.div_pcts = dpcts; ta.vec_s = vec_s;
        const size_t smem = ((size_t)4 * 64 * L.B + hadi_pb_mf_doubles(L.P) + (size_t)L.P * HADI_LC * HADI_PBW +
                             (dividend ? (size_t)(L.m1 + 2) + (size_t)8 * L.rowp : 0)) * sizeof(double) + 64;
        if (L.B == 8) emu::launch(8 * nb, 512, [&]() { hadi_team_kernel<8>(a, ta); }, smem, nb > 1);
        else emu::launch(8 * nb, 512, [&]() { hadi_team_kernel<4>(a, ta); }, smem, nb > 1);
        emu::launch(8, 64, [&]() { hadi_unpack_kernel(L, n_inst, dU.data(), U); });
        return g_err ? 4 : 0;
    }
    const size_t smem_small = american ? pl.smem_small_am : pl.smem_small_eu;
    if (use_small && !cs && !f32 && smem_small > 0) {
        HadiSmallArgs sm;
        sm.div_flag = dividend ? flags.data() : nullptr; sm.flag_stride = 0; sm.div_amounts = damounts; sm.div_pcts = dpcts;
        sm.vec_s = vec_s; sm.Nmax = N; sm.order = nullptr;
        if (use_small == 5 && !american && pl.L.nrows <= 32) {  // ... two instances per wavefront
            const size_t smem_seq = (size_t)hadi_small_seq_layout(pl.L.m1, pl.L.nrows).total * sizeof(double);
            if (pl.L.B == 1) emu::launch((n_inst + 1) / 2, 64, [&]() { hadi_small_seq2_kernel<1>(a, sm); }, 2 * smem_seq);
            else emu::launch((n_inst + 1) / 2, 64, [&]() { hadi_small_seq2_kernel<2>(a, sm); }, 2 * smem_seq);
        } else if (use_small == 3 && !american) {  // one wavefront per instance, sequential line solves (European / dividends)
            const size_t smem_seq = (size_t)hadi_small_seq_layout(pl.L.m1, pl.L.nrows).total * sizeof(double);
            if (pl.L.B == 1) emu::launch(n_inst, 64, [&]() { hadi_small_seq_kernel<1>(a, sm); }, smem_seq);
            else emu::launch(n_inst, 64, [&]() { hadi_small_seq_kernel<2>(a, sm); }, smem_seq);
        } else if (use_small == 2) {  // 8 wavefronts per instance (what hadi_api.hip picks for small batches)
            if (L.B == 1) {
                if (american) emu::launch(n_inst, 512, [&]() { hadi_small_kernel<1, 8, true>(a, sm); }, smem_small);
                else emu::launch(n_inst, 512, [&]() { hadi_small_kernel<1, 8, false>(a, sm); }, smem_small);
            } else {
                if (american) emu::launch(n_inst, 512, [&]() { hadi_small_kernel<2, 8, true>(a, sm); }, smem_small);
                else emu::launch(n_inst, 512, [&]() { hadi_small_kernel<2, 8, false>(a, sm); }, smem_small);
            }
        } else if (L.B == 1) {
            if (american) emu::launch(n_inst, 256, [&]() { hadi_small_kernel<1, 4, true>(a, sm); }, smem_small);
            else emu::launch(n_inst, 256, [&]() { hadi_small_kernel<1, 4, false>(a, sm); }, smem_small);
        } else {
            if (american) emu::launch(n_inst, 256, [&]() { hadi_small_kernel<2, 4, true>(a, sm); }, smem_small);
            else emu::launch(n_inst, 256, [&]() { hadi_small_kernel<2, 4, false>(a, sm); }, smem_small);
        }
        emu::launch(8, 64, [&]() { hadi_unpack_kernel(L, n_inst, dU.data(), U); });
        if (american && lam_out) emu::launch(8, 64, [&]() { hadi_unpack_kernel(L, n_inst, dLAM.data(), lam_out); });
        return 0;
    }
    if (pair_tab)  // (before any sweep of the streaming path, as hadi_api.hip does at the start of a sub-batch's time loop)
        emu::launch(pl.grid_as, 512, [&]() { hadi_pass_a_strip<8, 0, double, 2, 3>(a, 1); },
                    (size_t)4 * HADI_STRIP_NS(8, 2, 8) * pl.L.rowp * sizeof(double) + ((size_t)4 * 64 * 8 * 2 + (size_t)4 * 16) * sizeof(double));
    if (f32) {  // round the packed state to float, sweep on float arrays, widen again
        std::vector<float> fU(st), fY(st, 0.0f);
        emu::launch(8, 64, [&]() { hadi_narrow_kernel(L, dU.data(), fU.data(), st); });
        HadiSweepArgs af = a;
        af.U = reinterpret_cast<double *>(fU.data());
        af.Y = reinterpret_cast<double *>(fY.data());
        for (int n = 1; n <= N; n++)
            if (run_sweep_f32(pl, af, n)) return 2;
        emu::launch(8, 64, [&]() { hadi_widen_kernel(L, fU.data(), dU.data(), st); });
        emu::launch(8, 64, [&]() { hadi_unpack_kernel(L, n_inst, dU.data(), U); });
        return 0;
    }
    for (int n = 1; n <= N; n++) {
        const bool xstep = amp && (n == 1 || (dividend && flags[n - 1] >= 0));  // explicit (U, lambda_bar) step, as in hadi_api.hip
        if (xstep && n > 1)
            emu::launch(8, 64, [&]() { hadi_am_materialise_kernel(L, n_inst, ipar.data(), dU0.data(), dU.data(), dLAM.data(), pl.pos_m1); });
        if (amp && !xstep) {
            if (run_sweep_amp(pl, a, n)) return 2;
            continue;
        }
        if (dividend && flags[n - 1] >= 0) {  // device_solver.hpp:426-517 (host builds the step table, kernel applies)
            dUT = dU;
            emu::launch(8, 64, [&]() {
                hadi_dividend_kernel(L, n_inst, ipar.data(), vec_s, dUT.data(), dU.data(), flags.data(), 0, n, damounts, dpcts);
            });
        }
        if (run_row_pass(pl, a, n, cs ? 1 : 0)) return 2;
        run_col_pass(pl, cs ? av : a, n);
        if (cs) {
            run_row_pass(pl, av, n, 2);
            run_col_pass(pl, a, n);
        }
        if (xstep)
            emu::launch(8, 64, [&]() { hadi_am_dematerialise_kernel(L, n_inst, ipar.data(), dU0.data(), dU.data(), dLAM.data()); });
    }
    if (amp)
        emu::launch(8, 64, [&]() { hadi_am_materialise_kernel(L, n_inst, ipar.data(), dU0.data(), dU.data(), dLAM.data(), pl.pos_m1); });
    emu::launch(8, 64, [&]() { hadi_unpack_kernel(L, n_inst, dU.data(), U); });
    if (american && lam_out) emu::launch(8, 64, [&]() { hadi_unpack_kernel(L, n_inst, dLAM.data(), lam_out); });
    return 0;
}

// Tables only (serial host evaluation of hadi_setup_instance), for direct comparison with the oracle.
extern "C" int emu_tables(int m1, int m2, int N, double dt, double theta, double r_d, double r_f, double rho,
                          double sigma, double kappa, double eta, const double *vec_s, const double *vec_v,
                          const double *delta_s, const double *delta_v, int target_waves, double *scoef,
                          double *b2row, double *rowc, double *a2i, double *pb, double *rinv) {
    HadiPlan pl;
    if (hadi_make_plan(m1, m2, 1, target_waves, &pl, g_tune)) return 1;
    HadiSetupIn in;
    in.vec_s = vec_s; in.vec_v = vec_v; in.delta_s = delta_s; in.delta_v = delta_v;
    in.r_d = r_d; in.r_f = r_f; in.rho = rho; in.sigma = sigma; in.kappa = kappa; in.eta = eta;
    in.theta = theta; in.dt = dt; in.N = N;
    in.put = 0; in.strike = 0.0;
    std::vector<double> rwork(pl.n_rwork);
    HadiInstPar ip;
    HadiTables t;
    t.scoef = scoef; t.b2row = b2row; t.rowc = rowc; t.a2i = a2i; t.pb = pb; t.rinv = rinv;
    t.rwork = rwork.data(); t.ipar = &ip;
    hadi_setup_instance(pl.L, in, t, 0, 1, HadiNoSync());
    return 0;
}
