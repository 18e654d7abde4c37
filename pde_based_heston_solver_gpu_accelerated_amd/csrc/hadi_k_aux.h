// hadi_k_aux.h -- once-per-solve kernels: setup, pack / unpack, dividend jump, representation changes, fp32 narrowing, v-grid rebuild, price pick, LM partials.
// Part of libhadi's device code: include through hadi_kernels.h (which fixes the order).
#pragma once

// ------------------------------------------------------------------------------------------------
// Setup: one block per instance builds all operator tables (replaces bounds.initialize and the three
// build_matrix calls at the top of every reference launcher, e.g. jacobian_computation.cpp:255-261).
struct HadiSetupArgs {
    HadiLayout L;
    int n_inst;
    const double *vec_s, *vec_v, *delta_s, *delta_v;  // [n][..] natural arrays (device)
    const double *par;  // [n][8]: rho, sigma, kappa, eta, dt, N (as double), strike (put), option type (0 call, 1 put)
    double r_d, r_f, theta;
    double *scoef, *b2row, *rowc, *a2i, *pb, *rinv, *rwork;
    HadiInstPar *ipar;
};

struct HadiBlockSync {
    HADI_DEV void operator()() const { __syncthreads(); }
};

__global__ void __launch_bounds__(256) hadi_setup_kernel(HadiSetupArgs s) {
    const int inst = blockIdx.x;
    if (inst >= s.n_inst) return;
    const HadiLayout &L = s.L;
    HadiSetupIn in;
    in.vec_s = s.vec_s + (size_t)inst * (L.m1 + 1);
    in.vec_v = s.vec_v + (size_t)inst * (L.m2 + 1);
    in.delta_s = s.delta_s + (size_t)inst * L.m1;
    in.delta_v = s.delta_v + (size_t)inst * L.m2;
    const double *par = s.par + (size_t)inst * 8;
    in.rho = par[0]; in.sigma = par[1]; in.kappa = par[2]; in.eta = par[3];
    in.dt = par[4]; in.N = (int)par[5];
    in.r_d = s.r_d; in.r_f = s.r_f; in.theta = s.theta;
    in.strike = par[6]; in.put = (par[7] != 0.0) ? 1 : 0;
    HadiTables t;
    const int n4 = 4 * L.P;
    t.scoef = s.scoef + (size_t)inst * 4 * 64 * L.B * L.G;
    t.b2row = s.b2row + (size_t)inst * L.rowp;
    t.rowc = s.rowc + (size_t)inst * L.nrows * HADI_RC;
    t.a2i = s.a2i + (size_t)inst * 5 * L.nrows_pad;
    t.pb = s.pb + (size_t)inst * L.nrows_pad * HADI_PBW;
    t.rinv = s.rinv + (size_t)inst * n4 * n4;
    t.rwork = s.rwork + (size_t)inst * n4 * 2 * n4;
    t.ipar = s.ipar + inst;
    hadi_setup_instance(L, in, t, (int)threadIdx.x, (int)blockDim.x, HadiBlockSync());
}

// ------------------------------------------------------------------------------------------------
// Layout conversion natural [inst][j][i] <-> internal [inst][j][pos(i)] (pads written as 0).
// Instance k of the internal array reads natural instance k % n_src (a Jacobian batch replicates U_0).
__global__ void __launch_bounds__(256) hadi_pack_kernel(HadiLayout L, int n_inst, int n_src,
                                                        const double *__restrict__ nat, double *__restrict__ internal) {
    const size_t total = (size_t)n_inst * L.nrows_pad * L.rowp;
    const size_t m = (size_t)(L.m1 + 1) * L.nrows;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int slot = (int)(e % L.rowp);
        const size_t rowid = e / L.rowp;
        const int j = (int)(rowid % L.nrows_pad);
        const size_t inst = rowid / L.nrows_pad;
        const int i = hadi_slot_to_i(L, slot);
        double v = 0.0;
        if (i >= 0 && i <= L.m1 && j < L.nrows) v = nat[(inst % (size_t)n_src) * m + (size_t)j * (L.m1 + 1) + i];
        internal[e] = v;
    }
}

__global__ void __launch_bounds__(256) hadi_unpack_kernel(HadiLayout L, int n_inst, const double *__restrict__ internal,
                                                          double *__restrict__ nat) {
    const size_t m = (size_t)(L.m1 + 1) * L.nrows;
    const size_t total = (size_t)n_inst * m;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % (L.m1 + 1));
        const size_t rowid = e / (L.m1 + 1);
        const int j = (int)(rowid % L.nrows);
        const size_t inst = rowid / L.nrows;
        nat[e] = internal[inst * L.inst_stride + (size_t)j * L.rowp + hadi_pos(L, i)];
    }
}

__global__ void __launch_bounds__(256) hadi_fill_kernel(double *__restrict__ p, size_t n, double v) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) p[e] = v;
}

// ------------------------------------------------------------------------------------------------
// Discrete dividend jump (device_solver.hpp:448-504) on the internal layout: U <- interp(UT) where
// UT is a copy of U taken before the jump.  One thread per (row, s-node); the reference's linear
// search "first k with s_k > new_s" is a binary search on the ascending s-grid.  Which dividend (if any) an
// instance pays at the start of step n comes from the host-built table div_flag (see HadiSmallArgs).
__global__ void __launch_bounds__(256) hadi_dividend_kernel(HadiLayout L, int n_inst, const HadiInstPar *__restrict__ ipar,
                                                            const double *__restrict__ vec_s,
                                                            const double *__restrict__ UT, double *__restrict__ U,
                                                            const int *__restrict__ div_flag, int flag_stride, int n,
                                                            const double *__restrict__ div_amounts,
                                                            const double *__restrict__ div_pcts) {
    const int m1 = L.m1;
    const size_t per = (size_t)L.nrows * (m1 + 1);
    const size_t total = (size_t)n_inst * per;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % (m1 + 1));
        const size_t rowid = e / (m1 + 1);
        const int j = (int)(rowid % L.nrows);
        const size_t inst = rowid / L.nrows;
        const int dv = div_flag[inst * flag_stride + n - 1];
        if (dv < 0) continue;
        const double amount = div_amounts[dv], pct = div_pcts[dv];
        const double *__restrict__ s = vec_s + inst * (m1 + 1);
        const double *__restrict__ src = UT + inst * L.inst_stride + (size_t)j * L.rowp;
        const double old_s = s[i];
        const double new_s = old_s * (1.0 - pct) - amount;
        // ex-dividend spot <= 0: a call is worth 0 (device_solver.hpp:499-503), a put its s = 0 value
        double out = ipar[inst].put ? src[hadi_pos(L, 0)] : 0.0;
        if (new_s > 0) {
            // idx = first k in [0, m1] with s[k] > new_s, 0 if none
            int lo = 0, hi = m1 + 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s[mid] > new_s) hi = mid;
                else lo = mid + 1;
            }
            const int idx = (lo <= m1) ? lo : 0;
            if (idx > 0) {
                const double s_low = s[idx - 1], s_high = s[idx];
                const double weight = (new_s - s_low) / (s_high - s_low);
                const double val_low = src[hadi_pos(L, idx - 1)], val_high = src[hadi_pos(L, idx)];
                out = (1.0 - weight) * val_low + weight * val_high;
            } else {
                out = src[hadi_pos(L, 0)];
            }
        }
        U[inst * L.inst_stride + (size_t)j * L.rowp + hadi_pos(L, i)] = out;
    }
}

// ------------------------------------------------------------------------------------------------
// fp32-state sweep: the packed state is rounded to fp32 before the time loop and widened after it (same element layout).
__global__ void __launch_bounds__(256) hadi_narrow_kernel(HadiLayout L, const double *__restrict__ src, float *__restrict__ dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / L.rowp;
        const int x = (int)(e - row * L.rowp), i = hadi_slot_to_i(L, x);  // fp64 slot -> node -> fp32 slot (pads map to themselves)
        dst[row * L.rowp + (i >= 0 ? hadi_pos_f32(L.B, L.G, i) : x)] = (float)src[e];
    }
}
__global__ void __launch_bounds__(256) hadi_widen_kernel(HadiLayout L, const float *__restrict__ src, double *__restrict__ dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / L.rowp;
        const int x = (int)(e - row * L.rowp), i = hadi_slot_to_i(L, x);
        dst[e] = (double)src[row * L.rowp + (i >= 0 ? hadi_pos_f32(L.B, L.G, i) : x)];
    }
}

// ------------------------------------------------------------------------------------------------
// American, P representation <-> explicit (U, lambda_bar), elementwise on the packed arrays (payoff = its v-row 0):
//   materialise:    U = max(P, U0),  lambda_bar = max(0, (U0 - P)/dt)  (0 at i = m1)
//   dematerialise:  P = lambda_bar > 0 ? U0 - dt lambda_bar : U        (after a projection lambda_bar > 0 implies U = U0)
// Used for the first step (the caller's initial U need not dominate the payoff), around dividend steps (the jump acts
// on U alone) and for the outputs.
__global__ void __launch_bounds__(256) hadi_am_materialise_kernel(HadiLayout L, int n_inst, const HadiInstPar *__restrict__ ipar,
                                                                  const double *__restrict__ P0, double *__restrict__ UP,
                                                                  double *__restrict__ LAM, int pos_m1) {
    const size_t per = (size_t)L.nrows * L.rowp, total = (size_t)n_inst * per;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t inst = e / per, r = e - inst * per;
        const int x = (int)(r % L.rowp);
        const double pay = P0[inst * L.inst_stride + x], p = UP[inst * L.inst_stride + r];
        double lam = fmax(0.0, (pay - p) / ipar[inst].dt);
        if (x == pos_m1) lam = 0.0;
        UP[inst * L.inst_stride + r] = fmax(p, pay);
        LAM[inst * L.inst_stride + r] = lam;
    }
}
__global__ void __launch_bounds__(256) hadi_am_dematerialise_kernel(HadiLayout L, int n_inst, const HadiInstPar *__restrict__ ipar,
                                                                    const double *__restrict__ P0, double *__restrict__ UP,
                                                                    const double *__restrict__ LAM) {
    const size_t per = (size_t)L.nrows * L.rowp, total = (size_t)n_inst * per;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t inst = e / per, r = e - inst * per;
        const int x = (int)(r % L.rowp);
        const double lam = LAM[inst * L.inst_stride + r];
        if (lam > 0.0) UP[inst * L.inst_stride + r] = P0[inst * L.inst_stride + x] - ipar[inst].dt * lam;
    }
}

// ------------------------------------------------------------------------------------------------
// American payoff shape: mis[inst] != 0 if any v-row of the packed payoff differs from its row 0 (mis is zeroed first).
__global__ void __launch_bounds__(256) hadi_payoff_shape_kernel(HadiLayout L, int n_inst, const double *__restrict__ P0, int *__restrict__ mis) {
    const size_t per = (size_t)L.nrows * L.rowp, total = (size_t)n_inst * per;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t inst = e / per, r = e - inst * per;
        const size_t x = r % L.rowp;
        if (P0[inst * L.inst_stride + r] != P0[inst * L.inst_stride + x]) mis[inst] = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// GridViews::rebuild_variance_views (grid_pod.hpp:25-73) for every instance of the batch, each for its own V_0: one
// block per instance.  v_j = d sinh(j asinh(V/d)/m2), j = 0..m2; V_0 is pushed, the m2+2 values are sorted and the
// largest is dropped (the reference bubble-sorts them on one thread, grid_pod.hpp:47-57; the raw nodes are ascending,
// so sorting = inserting V_0 behind the last node <= V_0).  Delta_v follows.  sinh/asinh are the device library's, as
// in the reference's in-kernel rebuild.
__global__ void __launch_bounds__(256) hadi_rebuild_variance_kernel(int m2, int n_inst, const double *__restrict__ v0_i,
                                                                    double V, double d, double *__restrict__ vec_v,
                                                                    double *__restrict__ delta_v) {
    HADI_DYN_SMEM(double, raw);  // 2 (m2 + 1) doubles
    const int inst = blockIdx.x;
    if (inst >= n_inst) return;
    const int n = m2 + 1;
    double *outv = raw + n;
    const double V_0 = v0_i[inst];
    const double Delta_eta = (1.0 / m2) * asinh(V / d);
    for (int j = threadIdx.x; j < n; j += blockDim.x) raw[j] = d * sinh(j * Delta_eta);
    __syncthreads();
    int lo = 0, hi = n;  // pos = number of raw nodes <= V_0 (first index with raw > V_0)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (raw[mid] > V_0) hi = mid;
        else lo = mid + 1;
    }
    const int pos = lo;  // pos == n: V_0 is the largest of the m2+2 values and is the one dropped
    for (int k = threadIdx.x; k < n; k += blockDim.x) outv[k] = (k < pos) ? raw[k] : (k == pos) ? V_0 : raw[k - 1];
    __syncthreads();
    double *vv = vec_v + (size_t)inst * n, *dv = delta_v + (size_t)inst * m2;
    for (int k = threadIdx.x; k < n; k += blockDim.x) vv[k] = outv[k];
    for (int k = threadIdx.x; k < m2; k += blockDim.x) dv[k] = outv[k + 1] - outv[k];
}

// ------------------------------------------------------------------------------------------------
// Levenberg-Marquardt normal equations of this rank's rows on the device (replaces KokkosBlas::gemm("T","N") /
// gemv("T") and the residual kernel, jacobian_computation.cpp:117,154, heston_calibration.cpp:271-275):
//   out[0..24] = J^T J (row-major), out[25..29] = J^T r, out[30] = sum r^2,  r = market - model.
// One block, fixed-shape tree reduction: the result does not depend on scheduling (n is a few thousand at most).
__global__ void __launch_bounds__(256) hadi_lm_partials_kernel(int n, const double *__restrict__ J,
                                                               const double *__restrict__ model,
                                                               const double *__restrict__ market, double *__restrict__ out) {
    HADI_DYN_SMEM(double, redm);  // 21 x 256 doubles
    double (*red)[256] = reinterpret_cast<double (*)[256]>(redm);
    double acc[21];
#pragma unroll
    for (int q = 0; q < 21; q++) acc[q] = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) {
        double jr[5];
#pragma unroll
        for (int a = 0; a < 5; a++) jr[a] = J[(size_t)k * 5 + a];
        const double r = market[k] - model[k];
        int q = 0;
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int b = a; b < 5; b++) { acc[q] = fma(jr[a], jr[b], acc[q]); q++; }
#pragma unroll
        for (int a = 0; a < 5; a++) acc[15 + a] = fma(jr[a], r, acc[15 + a]);
        acc[20] = fma(r, r, acc[20]);
    }
#pragma unroll
    for (int q = 0; q < 21; q++) red[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int q = 0; q < 21; q++) red[q][threadIdx.x] += red[q][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int q = 0;
        for (int a = 0; a < 5; a++)
            for (int b = a; b < 5; b++) {
                out[a * 5 + b] = red[q][0];
                out[b * 5 + a] = red[q][0];
                q++;
            }
        for (int a = 0; a < 5; a++) out[25 + a] = red[15 + a][0];
        out[30] = red[20][0];
    }
}

// ------------------------------------------------------------------------------------------------
// Price pick (jacobian_computation.cpp:275-288): first s-node with |s_i - S_0| < 1e-10, first
// v-node with |v_j - V_0| < 1e-10 (0 if none, grid_pod.hpp:76-87).  status[inst] = 1 if S_0 is off-grid.
__global__ void __launch_bounds__(64) hadi_pick_kernel(HadiLayout L, int n_inst, const double *__restrict__ vec_s,
                                                       const double *__restrict__ vec_v, const double *__restrict__ U,
                                                       double S_0, const double *__restrict__ V0_i, double V_0,
                                                       double *__restrict__ prices, int price_stride,
                                                       int *__restrict__ status) {
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= n_inst) return;
    const double *s = vec_s + (size_t)inst * (L.m1 + 1);
    const double *v = vec_v + (size_t)inst * (L.m2 + 1);
    const double v0 = V0_i ? V0_i[inst] : V_0;
    int is = -1, iv = 0;
    for (int i = 0; i <= L.m1; i++)
        if (fabs(s[i] - S_0) < 1e-10) { is = i; break; }
    for (int j = 0; j <= L.m2; j++)
        if (fabs(v[j] - v0) < 1e-10) { iv = j; break; }
    if (is < 0) {
        status[inst] = 1;
        prices[(size_t)inst * price_stride] = nan("");
        return;
    }
    status[inst] = 0;
    prices[(size_t)inst * price_stride] = U[(size_t)inst * L.inst_stride + (size_t)iv * L.rowp + hadi_pos(L, is)];
}

// J(k, param) = (perturbed price - base price) / eps from the 6 n0 prices of a flattened Jacobian sweep (groups: base, kappa,
// eta, sigma, rho, v0), jacobian_computation.cpp:329,360.
__global__ void __launch_bounds__(256) hadi_jacobian_rows_kernel(int n0, const double *__restrict__ prices, double eps,
                                                                 double *__restrict__ J, double *__restrict__ base) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n0) return;
    const double b = prices[k];
    base[k] = b;
    for (int g = 1; g <= 5; g++) J[(size_t)k * 5 + (g - 1)] = (prices[(size_t)g * n0 + k] - b) / eps;
}

// Diagnostics (hadi_debug_rcp): the reciprocal every line solve of the sweep uses, elementwise.
__global__ void __launch_bounds__(256) hadi_rcp_kernel(int n, const double *__restrict__ x, double *__restrict__ out) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) out[e] = hadi_rcp(x[e]);
}

// Replicate one of `nsrc` source rows (length len) into every instance's row: dst[inst] = src[sel[inst]]
// (sel == nullptr: src row 0).  Used to hand every instance the v-grid rebuilt for V_0 (or V_0+eps).
__global__ void __launch_bounds__(256) hadi_bcast_rows_kernel(int len, int n_inst, const double *__restrict__ src,
                                                              const int *__restrict__ sel, double *__restrict__ dst) {
    const size_t total = (size_t)n_inst * len;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t inst = e / len;
        const int k = (int)(e - inst * len);
        const int sr = sel ? sel[inst] : 0;
        dst[e] = src[(size_t)sr * len + k];
    }
}
