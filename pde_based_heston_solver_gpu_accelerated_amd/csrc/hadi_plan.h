// hadi_plan.h -- host-side choice of layout and launch geometry for one batch shape.
#pragma once
#include <stdlib.h>

#include "hadi_core.h"

// hadi_pass_b2: rows of the next tile each wavefront prefetches into LDS (16 chunks: 96 KB for the reduced system -- exchange
// values, selected inverse rows, their product -- + 16 x NPF x 64 elements = the CU's whole 160 KB)
#define HADI_B2_NPF(ES) (HADI_PB_MF ? ((ES) == 4 ? 16 : 8) : ((ES) == 4 ? 24 : 12))

struct HadiPlan {
    HadiLayout L;
    int W;               // pass A: v-rows solved concurrently by one tile group (W*G wavefronts)
    int NG, PD;          // pass A: tile groups per block, prefetch depth in iterations
    int R, ntiles;       // pass A: v-rows per block tile, tiles per instance
    int grid_a;          // pass A grid (64*W-thread blocks), padded to a multiple of 8 for the XCD remap
    size_t smem_a;       // pass A dynamic LDS bytes: NG rings of (PD+1)W+4 rows + the 4 s-coefficient arrays + tables
    // strip row pass (hadi_pass_a_strip; 2, 4 or 8 nodes per lane, one wavefront per row): v-rows per wavefront strip,
    // 8-strip blocks per instance, grid, LDS bytes; use_strip = 0 keeps the shared-ring kernel
    int use_strip, RS, sblocks, grid_as;
    size_t smem_as;
    int ctiles;          // pass B: 64-column tiles per instance
    int btpw, bgroups;   // pass B: column tiles per block (register double-buffered), blocks per instance
    int grid_b, block_b; // pass B grid / block (P*64 threads)
    size_t smem_b;       // pass B dynamic LDS bytes
    // 9 .. 16 chunks, European: hadi_pass_b2 -- one exchange buffer + each wavefront's prefetch area of HADI_B2_NPF(ES) rows
    size_t smem_b2;
    int pos_m1;
    // table sizes per instance (doubles)
    size_t n_scoef, n_b2row, n_rowc, n_a2i, n_pb, n_rinv, n_rwork;
    // small-grid path: whole instance in LDS, one launch for the whole time loop (0 = not applicable)
    size_t smem_small_eu, smem_small_am;
    // Shapes beyond the streaming kernels: more than 1024 s-intervals -> sequential row pass (hadi_pass_a_seq, one lane per
    // v-row); more than HADI_MAX_P * HADI_LC v-rows -> sequential column pass (hadi_pass_b_seq, one lane per column)
    int row_seq, col_seq;
    // 128 < m1 <= 256: the strips run two per wavefront (hadi_pass_a_pairs); RS / sblocks / grid_as then describe THAT geometry
    // (8 strips per 4-wavefront block)
    int use_pairs;
    size_t smem_pairs_eu, smem_pairs_amp;
};

// Execution-path choices a caller may override through hadi_set_tuning (tests force kernel variants with them; results
// agree to round-off).  Defaults = automatic.  There is no environment-variable back door.
struct HadiTuning {
    int row_tile = 0;     // shared-ring row pass: v-rows per block tile (0 = automatic)
    int strip = -1;       // strip row pass: -1 automatic, 0 never, 1 whenever the geometry allows it
    int col_groups = 0;   // column pass: blocks per instance (0 = automatic)
    int small_waves = 0;  // LDS-resident small-grid kernel: wavefronts per instance, 4 or 8 (0 = automatic)
    int strip_blocks = 0; // strip row pass: blocks per instance (0 = automatic); the strips get ceil(rows / (strips per block x blocks)) rows
    int pair_strips = -1; // 4 nodes per lane: two strips per wavefront on the 8-node arithmetic (hadi_pass_a_pairs): -1 automatic, 0 never, 1 always
    // Constants of the strips-or-ring cost model below, measured on one MI355X (the boxes of a pool differ by +-3 % on the very
    // kernels they model, and the crossover sits inside that band): adjustable per handle (hadi_set_tuning "model_*")
    int strip_row_ns = 2800, ring_row_ps = 2330, ring_fixed_ns = 12000;       // 8 nodes per lane, one wavefront per row
    int pstrip_row_ns = 3250, pring_row_ps = 4300, pring_fixed_ns = 15000;    // two wavefronts per row
};

// Returns 0 on success, 1 if the shape is outside what the kernels cover.
// state_bytes: element size of the state arrays the sweep streams (8; 4 for the fp32-state sweep) -- it decides the row pitch.
inline int hadi_make_plan(int m1, int m2, int n_inst, int target_waves, HadiPlan *out, const HadiTuning &tu = HadiTuning(),
                          int state_bytes = 8) {
    if (m1 < 2 || m2 < 3 || n_inst < 1) return 1;
    if (m1 > (1 << 20) || m2 > (1 << 20)) return 1;  // (index arithmetic in 32 bits; memory runs out long before)
    HadiPlan p;
    HadiLayout &L = p.L;
    L.m1 = m1; L.m2 = m2; L.nrows = m2 + 1;
    hadi_pick_shape(m1, &L.B, &L.G);
    L.rowp = 64 * L.B * L.G + HADI_ROW_PAD(L.B, state_bytes);
    L.P = (L.nrows + HADI_LC - 1) / HADI_LC;
    L.lc = HADI_LC;
    p.row_seq = (m1 > 1024) ? 1 : 0;
    p.col_seq = 0;
    if (L.P > HADI_MAX_P) {  // one chunk of all rows: the column pass sweeps each column sequentially (hadi_pass_b_seq)
        p.col_seq = 1;
        L.P = 1;
        L.lc = L.nrows;
    }
    L.nrows_pad = L.P * L.lc;
    L.inst_stride = (long long)L.rowp * L.nrows_pad;
    if (L.inst_stride * 8 >= (1ll << 31)) return 1;  // (the column pass addresses an instance through a 32-bit buffer offset)
    // Row tiles: as tall as possible (halo re-reads cost 4/R) while the launch still has a few blocks per CU
    // (target_waves = 8 per CU), a multiple of W rows each.
    p.W = 4;
    const int W = p.W;
    // One tile group per block and one iteration of prefetch at 8 nodes per lane (two 4-wave blocks per CU).
    // Measured on MI355X at m1 = 512: NG = 2 groups sharing a block with PD = 2 (one 8-wave block per CU,
    // 158 KB LDS) runs 0.170 ms/launch against 0.159 -- the wider barrier costs more than the deeper
    // prefetch gains.  Narrower rows have LDS to spare and prefetch two iterations ahead.
    p.NG = 1;
    p.PD = (L.B == 8) ? 1 : 2;
    const int Rmax = (p.NG == 2) ? 44 : 64;  // LDS: NG compact row tables of R rows
    int ntiles = ((3 * target_waves) / 8 + n_inst - 1) / n_inst;
    if (ntiles < 1) ntiles = 1;
    int R = (L.nrows + ntiles - 1) / ntiles;
    R = (R + W - 1) / W * W;
    if (R < W) R = W;
    if (R > Rmax) R = Rmax;
    ntiles = (L.nrows + R - 1) / R;
    if (p.NG == 2 && (ntiles & 1) && ntiles > 1) ntiles++;  // pair the tiles up
    R = ((L.nrows + ntiles - 1) / ntiles + W - 1) / W * W;  // balance
    ntiles = (L.nrows + R - 1) / R;
    if (tu.row_tile > 0) { R = tu.row_tile; if (R < W) R = W; R = (R + W - 1) / W * W; if (R > Rmax) R = Rmax; ntiles = (L.nrows + R - 1) / R; }
    p.R = R;
    p.ntiles = ntiles;
    p.smem_a = ((size_t)p.NG * ((p.PD + 1) * W + 4) * L.rowp + (size_t)4 * 64 * L.B * L.G + (size_t)p.NG * 8 * W +
                (size_t)p.NG * R * HADI_RCL) * sizeof(double);
    const long long total = (long long)n_inst * ((ntiles + p.NG - 1) / p.NG);
    p.grid_a = (int)((total + 7) / 8 * 8);
    // Strip row pass: every wavefront walks down RS consecutive v-rows alone.  Strips shorter than 16 rows re-read too
    // many halo rows (4 per strip), so small batches stay on the shared-ring kernel.
    p.use_strip = 0; p.RS = 0; p.sblocks = 0; p.grid_as = 0; p.smem_as = 0;
    if (L.B >= 2 && L.G == 1) {
        long long rs0 = ((long long)L.nrows * n_inst + target_waves - 1) / target_waves;
        if (rs0 < 16) rs0 = 16;
        if (rs0 > 64) rs0 = 64;
        const int ns = (L.nrows + (int)rs0 - 1) / (int)rs0;
        const int nwv = HADI_STRIP_WAVES(L.B);
        p.sblocks = (ns + nwv - 1) / nwv;
        p.RS = (L.nrows + nwv * p.sblocks - 1) / (nwv * p.sblocks);
        p.grid_as = (int)(((long long)n_inst * p.sblocks + 7) / 8 * 8);
        p.smem_as = ((size_t)nwv * HADI_STRIP_NS(L.B, 1, 8) * L.rowp + (size_t)4 * 64 * L.B) * sizeof(double);
        // hadi_set_tuning("strip", 1) forces strips wherever the geometry allows them (tests).
        const int cus = target_waves / 8 > 0 ? target_waves / 8 : 1;
        p.use_strip = 0;
        if (L.B == 8) {
            // One strip block occupies a CU, so the launch runs in ceil(blocks / CUs) ROUNDS of RS row steps each plus a
            // fixed cost per round (prologue, 4 halo rows, tail): measured 0.110 / 0.065 / 0.042 ms per round at RS = 33 /
            // 17 / 11, i.e. ~ (RS + 6) x 2.8 us.  Pick the number of blocks per instance that minimises rounds x (RS + 6):
            // 64 instances -> 3 blocks of 11-row strips (0.053 -> 0.042 ms per launch), 160 -> 3 blocks (2 rounds instead
            // of a 320-block launch whose second round is a quarter full), 256 -> 1 block of 33-row strips.  The shared
            // ring (small blocks, ~2.3 ns per row of the batch + 12 us) keeps the batches too small to fill the CUs.
            int best_sb = 0;
            double best_cost = 0.0;
            for (int sb = 1; sb <= 6; sb++) {
                const int rs = (L.nrows + nwv * sb - 1) / (nwv * sb);
                if (rs < 8 || rs > 64) continue;
                const long long blocks = (long long)n_inst * sb, rounds = (blocks + cus - 1) / cus;
                const double cost = (double)rounds * (rs + 6);
                if (!best_sb || cost < best_cost - 1e-9) { best_sb = sb; best_cost = cost; }
            }
            if (best_sb) {
                const double t_strip = best_cost * tu.strip_row_ns * 1e-6, t_ring = tu.ring_row_ps * 1e-9 * (double)n_inst * L.nrows + tu.ring_fixed_ns * 1e-6;  // ms
                p.sblocks = best_sb;
                p.RS = (L.nrows + nwv * best_sb - 1) / (nwv * best_sb);
                p.grid_as = (int)(((long long)n_inst * p.sblocks + 7) / 8 * 8);
                p.use_strip = (t_strip < t_ring) ? 1 : 0;
            }
        }
        const long long sblk = (long long)n_inst * p.sblocks;
        // 2 nodes per lane (64 < m1 <= 128), batches of several blocks per CU: 4-strip blocks beat the shared ring
        // (128x64 x2000: 0.130 -> 0.108 ms per launch); at 4 nodes per lane the two are level (0.165 vs 0.167).
        if (L.B == 2 && p.RS >= 16 && p.RS <= 64 && sblk >= 4 * (long long)cus) p.use_strip = 1;
        // 4 nodes per lane (128 < m1 <= 256): with the unrolled LDS-DMA fetch the strips are level with the shared ring
        // at 1024 instances (0.159 ms per launch both) and ahead below that (256x128 x300: 0.063 -> 0.053; 200x100 x700:
        // 0.094 -> 0.076; 512 American puts in the P representation: 0.094 -> 0.084)
        if (L.B == 4 && p.RS >= 16 && p.RS <= 64 && sblk >= (long long)cus) p.use_strip = 1;
        if (tu.strip >= 0) p.use_strip = (tu.strip && p.RS >= 1 && p.RS <= 64) ? 1 : 0;
    }
    // Paired strips (512 < m1 <= 1024, hadi_pass_a_strip<8, EU, T, 2>): 4 pairs of wavefronts per block, one strip per pair.
    // Same round model as above; a paired row step costs more than a single-wavefront one (second right-hand side in the
    // cyclic reduction, the pair rendezvous).  The caller keeps American and Craig-Sneyd sweeps on the shared ring.
    if (L.B == 8 && L.G == 2) {
        const int spb = HADI_STRIP_WAVES(L.B) / 2, ns = state_bytes == 8 ? 3 : 4;
        const int cus = target_waves / 8 > 0 ? target_waves / 8 : 1;
        int best_sb = 0;
        double best_cost = 0.0;
        for (int sb = 1; sb <= 8; sb++) {
            const int rs = (L.nrows + spb * sb - 1) / (spb * sb);
            if (rs < 8 || rs > 64) continue;
            const long long blocks = (long long)n_inst * sb, rounds = (blocks + cus - 1) / cus;
            const double cost = (double)rounds * (rs + 6);
            if (!best_sb || cost < best_cost - 1e-9) { best_sb = sb; best_cost = cost; }
        }
        if (!best_sb && tu.strip == 1) { best_sb = 1; best_cost = 1e30; }  // forced (tests): one block of four short strips
        if (best_sb) {
            // measured at 1024x512 (fp64 state), ms per launch, strips / ring: 8 instances 0.067 / 0.034, 16: 0.068 / 0.051,
            // 32: 0.071 / 0.085, 64: 0.127 / 0.143, 128: 0.254 / 0.309, 256: 0.498 / 0.562
            const double t_strip = best_cost * tu.pstrip_row_ns * 1e-6, t_ring = tu.pring_row_ps * 1e-9 * (double)n_inst * L.nrows + tu.pring_fixed_ns * 1e-6;  // ms
            p.sblocks = best_sb;
            p.RS = (L.nrows + spb * best_sb - 1) / (spb * best_sb);
            p.grid_as = (int)(((long long)n_inst * p.sblocks + 7) / 8 * 8);
            p.smem_as = (size_t)spb * ns * L.rowp * state_bytes + ((size_t)4 * 64 * L.B * L.G + (size_t)spb * 16) * sizeof(double);
            p.use_strip = (t_strip < t_ring) ? 1 : 0;
            if (tu.strip >= 0) p.use_strip = tu.strip ? 1 : 0;
        }
    }
    // Pair strips at 4 nodes per lane: wherever the strips are chosen (or forced), unless the caller keeps the plain ones.
    p.use_pairs = 0;
    p.smem_pairs_eu = ((size_t)4 * 4 * 544 + 4 * 256 + 32) * sizeof(double);
    p.smem_pairs_amp = ((size_t)4 * 4 * 544 + 4 * 256 + 272 + 32) * sizeof(double);
    if (L.B == 4 && L.G == 1 && p.use_strip && tu.pair_strips != 0 && state_bytes == 8) {
        // 8 strips per block; as many blocks per instance as keep the strips at 16 rows or more, while the launch still offers
        // two blocks per CU
        const int cus = target_waves / 8 > 0 ? target_waves / 8 : 1;
        int sb = 1;
        while ((long long)n_inst * sb < 2ll * cus && (L.nrows + 8 * (sb + 1) - 1) / (8 * (sb + 1)) >= 16) sb++;
        const int rs = (L.nrows + 8 * sb - 1) / (8 * sb);
        // A wavefront of pairs does the work of two, so the launch has half the wavefronts of the plain strips: it pays
        // (measured, ms per launch pairs / plain: 256x128 x512 American P 0.079 / 0.0815, x1024 0.153 / 0.159, x1024 European
        // 0.138 / 0.148) only where two 4-wavefront blocks per CU are still there and the strips keep 16 rows -- below that the
        // plain strips win (x256: 0.059 / 0.047, 200x100 x700 with 13-row strips: 0.094 / 0.074).
        if (tu.pair_strips == 1 || (rs >= 16 && (long long)n_inst * sb >= 2ll * cus)) {
            p.use_pairs = 1;
            p.sblocks = sb;
            p.RS = rs;
            p.grid_as = (int)(((long long)n_inst * sb + 7) / 8 * 8);
        }
    }
    if (tu.strip_blocks > 0 && L.B >= 2 && p.use_pairs) {
        const int rs = (L.nrows + 8 * tu.strip_blocks - 1) / (8 * tu.strip_blocks);
        if (rs >= 1) { p.sblocks = tu.strip_blocks; p.RS = rs; p.grid_as = (int)(((long long)n_inst * p.sblocks + 7) / 8 * 8); }
    } else
    if (tu.strip_blocks > 0 && L.B >= 2) {  // forced geometry (measurements): blocks per instance
        const int spb = (L.G == 2) ? HADI_STRIP_WAVES(L.B) / 2 : HADI_STRIP_WAVES(L.B);
        const int rs = (L.nrows + spb * tu.strip_blocks - 1) / (spb * tu.strip_blocks);
        if (rs >= 1 && rs <= 64 && p.smem_as > 0) {
            p.sblocks = tu.strip_blocks;
            p.RS = rs;
            p.grid_as = (int)(((long long)n_inst * p.sblocks + 7) / 8 * 8);
        }
    }
    p.ctiles = (L.rowp + 63) / 64;
    // Column tiles per block.  The pitch is 64*B*G + pad, so an instance has nfull = B*G full tiles and one SHORT tile
    // (the pad columns: little traffic, but a whole solve -- about 0.3 of a full tile's time).  The full tiles are dealt
    // out btpw per block and the short tile rides with the last block (hadi_pb_tile_range).  A block keeps a CU to itself
    // above four chunks, so a launch takes ceil(blocks / CUs) rounds of its heaviest block; blocks of a single tile lose
    // the register double-buffering (+30 % measured, hence the 0.6).
    {
        const int nfull = L.rowp / 64, nshort = p.ctiles - nfull;
        const int cus = target_waves / 8 > 0 ? target_waves / 8 : 1;
        auto gfull_of = [&](int b) { return nfull > 0 ? (nfull + b - 1) / b : 1; };
        // own = the short tile gets a block of its own instead of riding with the last block
        auto cost = [&](int b, bool own) {
            const int gf = gfull_of(b);
            // up to eight chunks the kernel holds three tiles in registers, all loaded up front; a fourth tile waits for a
            // free buffer (512x256 x256: 2 blocks of 4 and 4+short tiles 0.129 ms against 0.109 for 3 blocks of 3,3,2+short)
            // (a single FOURTH tile is free when the blocks do not queue up behind each other on a CU: in a single-round launch
            // its wait overlaps the other CUs' traffic -- 512x256 x128: 2 blocks of 4 and 4+short tiles 0.0628 ms against 0.0663
            // for 4 blocks of 2)
            const bool queued = (long long)n_inst * (gf + (own ? 1 : 0)) > cus;
            auto weight = [&](double tiles, int count) {
                return tiles + (count < 2 ? 0.6 : 0.0) + ((L.P <= 8 && count > 3 && (queued || count > 4)) ? 1.0 * (count - 3) : 0.0);
            };
            const int last_full = nfull - (gf - 1) * b;
            // a launch that leaves half the CUs idle is bound by the latency of its longest block, not by traffic: there the
            // short tile costs a whole tile time (512x256, ONE instance: 17.5 -> 24.8 ms per 1000 steps with it appended)
            const double sw = (2ll * n_inst * (gf + (own ? 1 : 0)) <= cus) ? 1.0 : 0.3;
            double w = gf > 1 ? weight((double)b, b) : 0.0;
            if (own) {
                const double wl = weight((double)last_full, last_full), ws = weight(sw, 1);
                if (wl > w) w = wl;
                if (ws > w) w = ws;
            } else {
                // (the single-tile penalty looks at all tiles of the block, the late-tile penalty at its FULL tiles: the short
                // one behind them is a sliver of traffic)
                double wl = weight(last_full + sw * nshort, last_full);
                if (last_full + nshort >= 2 && last_full < 2) wl -= 0.6;
                if (wl > w) w = wl;
            }
            const long long rounds = ((long long)n_inst * (gf + (own ? 1 : 0)) + cus - 1) / cus;
            return (double)rounds * w;
        };
        // start from "about three blocks per CU in the launch" and move only to a split that needs fewer tile-rounds
        const int want_blocks = (3 * target_waves) / 8;  // target_waves = 8 per CU
        int groups = (want_blocks + n_inst - 1) / n_inst;
        if (groups < 1) groups = 1;
        if (groups > p.ctiles) groups = p.ctiles;
        int btpw = (p.ctiles + groups - 1) / groups;
        if (btpw > nfull && nfull > 0) btpw = nfull;
        bool own_short = false;
        for (int b = 1; b <= nfull; b++)
            for (int own = 0; own <= (nshort ? 1 : 0); own++)
                if (cost(b, own != 0) < cost(btpw, own_short) - 1e-9) { btpw = b; own_short = own != 0; }
        // Up to four chunks (m2 <= 131) a block is at most 256 threads and two or more of them share a CU: blocks of ONE
        // tile then overlap each other better than tiles pipeline inside a block (measured, 256x128 x512 American:
        // 0.094 -> 0.083 ms per launch; 200x100 x700: 0.087 -> 0.084; 128x64 x2000: 0.070 -> 0.068); the short tile gets
        // a block of its own there.
        if (L.P <= 4) { btpw = 1; own_short = nshort > 0; }
        if (tu.col_groups > 0) {
            int g = tu.col_groups < p.ctiles ? tu.col_groups : p.ctiles;
            own_short = (g == p.ctiles && nshort > 0 && nfull > 0);
            if (own_short) g = nfull;
            btpw = nfull > 0 ? (nfull + g - 1) / g : 1;
        }
        if (btpw < 1) btpw = 1;
        p.btpw = btpw;
        p.bgroups = gfull_of(btpw) + (own_short ? 1 : 0);
    }
    p.grid_b = (n_inst * p.bgroups + 7) / 8 * 8;  // padded to a multiple of 8 for the XCD remap
    p.block_b = 64 * L.P;
    // the reduced system on the matrix core: exchange values, the selected inverse rows (transposed) and their product; the
    // A/B build without it: two interface-exchange buffers + each wavefront's four rows of the reduced inverse
    p.smem_b = HADI_PB_MF ? hadi_pb_mf_doubles(L.P) * sizeof(double) : (size_t)L.P * (2 * 4 * 64 + 16 * L.P) * sizeof(double);
    p.smem_b2 = (HADI_PB_MF ? hadi_pb_mf_doubles(L.P) : (size_t)L.P * (4 * 64 + 16 * L.P)) * sizeof(double) +
                (size_t)L.P * HADI_B2_NPF(state_bytes) * 64 * state_bytes;
    p.pos_m1 = hadi_pos(L, m1);
    p.n_scoef = (size_t)4 * 64 * L.B * L.G;
    p.n_b2row = (size_t)L.rowp;
    p.n_rowc = (size_t)L.nrows * HADI_RC;
    p.n_a2i = (size_t)5 * L.nrows_pad;
    p.n_pb = (size_t)L.nrows_pad * HADI_PBW;
    p.n_rinv = (size_t)16 * L.P * L.P;
    p.n_rwork = (size_t)32 * L.P * L.P;
    p.smem_small_eu = p.smem_small_am = 0;
    if (L.G == 1 && L.B <= 2 && L.P == 1) {
        const size_t fixed = (size_t)(L.nrows + 4) * L.rowp + (size_t)L.nrows * L.rowp + (size_t)4 * 64 * L.B +
                             (size_t)L.nrows * HADI_RCL + (size_t)L.nrows * HADI_PBW;
        const size_t eu = fixed * sizeof(double), am = (fixed + (size_t)2 * L.nrows * L.rowp) * sizeof(double);
        if (eu <= 76 * 1024) p.smem_small_eu = eu;   // two blocks per CU
        if (am <= 150 * 1024) p.smem_small_am = am;
    }
    *out = p;
    return 0;
}

// Fraction of the row pass's CU-rounds that a STRIP launch of this plan leaves idle: blocks / (rounds x block slots of the chip).
// A strip block of 8 nodes per lane holds a CU alone, the 4-node strips and the pair strips share it with a second block, the
// 2-node strips with three more (their launch bounds): there a launch really runs in rounds of equal blocks, and this number
// is what decides between one and two streams (hadi_api.hip, run_sweep).  A batch whose row pass fills whole rounds (512x256:
// 64, 128, 256, 512 instances) gains nothing from a second stream and loses 2 - 13 % to two kernels that each want every CU;
// one that leaves a partial round idle (512x256: 96 and 192 instances 25 %, 160: 6 %; 1024x512 x96: 25 %; 256x128 x700
// American: 18 %) gains 2 - 23 % when its two halves run side by side, the row pass of one on the CUs the other's partial
// round leaves free (gpurun_out/r04b/stream_sweep.log).  The shared-ring row pass (small blocks, three
// or four per CU, the chip refilled block by block) shows no such pattern -- of 18 measured cases 13 gain up to 24 % and 5
// lose up to 10 %, with nothing in the launch geometry that tells them apart -- and stays on one stream (0 here).
// Deterministic per (shape, batch size): the path of a call never depends on the calls before it.
inline double hadi_plan_row_idle(const HadiPlan &p, int n_inst, int cus) {
    if (p.row_seq || !p.use_strip) return 0.0;
    const int per_cu = (p.L.B == 8) ? 1 : (p.use_pairs || p.L.B == 4) ? 2 : 4;
    const long long blocks = (long long)n_inst * p.sblocks, slots = (long long)cus * per_cu;
    if (blocks < 1 || slots < 1) return 0.0;
    const long long rounds = (blocks + slots - 1) / slots;
    return 1.0 - (double)blocks / (double)(rounds * slots);
}
#define HADI_TWO_STREAM_IDLE 0.04  // two streams from this idle fraction on (160 instances of 512x256: 0.0625, +7 % measured)

// Which dividend an instance with (N, dt) pays at the START of step n = 1..N (device_solver.hpp:426-447,508-516):
// flags[n-1] = index into the schedule or -1.  n*dt is evaluated in floating point exactly as the reference
// does (12*0.05 = 0.6000000000000001 decides the step a dividend lands on).  flags has `len` >= N entries;
// entries beyond N are -1.
inline void hadi_dividend_steps(int N, double dt, int num_div, const double *dates, int *flags, int len) {
    int cur = 0;
    for (int n = 1; n <= len; n++) {
        flags[n - 1] = -1;
        if (n > N) continue;
        const double t = n * dt;
        if (cur < num_div && t <= dates[cur] && dates[cur] < (n + 1) * dt) flags[n - 1] = cur;
        if (cur < num_div && t > dates[cur]) cur++;
    }
}
