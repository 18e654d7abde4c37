// hadi_host.hpp -- C++ host mirror of the reference's launcher interface on top of the C ABI (include/hadi.h).
//
// The reference (BCW-dot/PDE-based-Heston-Solver-GPU-accelerated) is C++ on Kokkos; Kokkos is not available in this
// image, so this header offers the same host-callable entry points -- same names, same argument order and meaning --
// over std::vector / raw pointers instead of Kokkos::View, header-only, no dependency besides libhadi:
//
//   Grid                                   Grid::Grid                        src/grid.cpp:16-61
//   GridViews / buildMultipleGridViews     grid_pod.hpp:8-111                (the four arrays of a batch, [n][len])
//   DO_Workspace                           DO_solver_workspace.hpp:4-44      (only U and lambda_bar cross the boundary)
//   parallel_DO_solve                      device_solver.hpp:52-185
//   compute_base_prices{,_american,_dividends,_american_dividends}   jacobian_computation.cpp:368, 629, 922, 1232
//   compute_jacobian{,_american,_dividends,_american_dividends}      jacobian_computation.cpp:204, 457, 726, 1031
//   compute_parameter_update_on_device                               jacobian_computation.cpp:107-195
//
// Arguments of the reference that only carried scratch (A0/A1/A2 solver arrays, bounds_d) do not exist here; failures
// throw std::runtime_error with the library's message (the reference's launchers return void).  With Kokkos present a
// maintainer binds View::data() pointers instead -- INTEGRATION.md shows that shim.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "hadi.h"

namespace hadi_host {

struct Handle {  // one per process and GPU, next to Kokkos::initialize in the reference's main()
    hadi_ctx *ctx = nullptr;
    explicit Handle(int device = 0) {
        const int rc = hadi_create(&ctx, device);
        if (rc) throw std::runtime_error(std::string("hadi_create: ") + hadi_status_string(rc));
    }
    ~Handle() { hadi_destroy(ctx); }
    Handle(const Handle &) = delete;
    Handle &operator=(const Handle &) = delete;
};

// src/grid.cpp:16-61 (same constructor arguments)
struct Grid {
    int m1, m2;
    std::vector<double> Vec_s, Vec_v, Delta_s, Delta_v;
    Grid(int m1_, double S, double S_0, double K, double c, int m2_, double V, double V_0, double d)
        : m1(m1_), m2(m2_), Vec_s(m1_ + 1), Vec_v(m2_ + 1), Delta_s(m1_), Delta_v(m2_) {
        if (hadi_make_grid(m1, S, S_0, K, c, m2, V, V_0, d, Vec_s.data(), Vec_v.data(), Delta_s.data(), Delta_v.data()))
            throw std::runtime_error("hadi_make_grid failed");
    }
};

// grid_pod.hpp:8-111: the GridViews of a batch as four [nInstances][len] host arrays
struct GridViews {
    int nInstances = 0, m1 = 0, m2 = 0;
    std::vector<double> Vec_s, Vec_v, Delta_s, Delta_v;
    void set(int i, const Grid &g) {
        std::copy(g.Vec_s.begin(), g.Vec_s.end(), Vec_s.begin() + (size_t)i * (m1 + 1));
        std::copy(g.Vec_v.begin(), g.Vec_v.end(), Vec_v.begin() + (size_t)i * (m2 + 1));
        std::copy(g.Delta_s.begin(), g.Delta_s.end(), Delta_s.begin() + (size_t)i * m1);
        std::copy(g.Delta_v.begin(), g.Delta_v.end(), Delta_v.begin() + (size_t)i * m2);
    }
};
inline void buildMultipleGridViews(GridViews &grids, int nInstances, int m1, int m2) {
    grids.nInstances = nInstances; grids.m1 = m1; grids.m2 = m2;
    grids.Vec_s.assign((size_t)nInstances * (m1 + 1), 0.0);
    grids.Vec_v.assign((size_t)nInstances * (m2 + 1), 0.0);
    grids.Delta_s.assign((size_t)nInstances * m1, 0.0);
    grids.Delta_v.assign((size_t)nInstances * m2, 0.0);
}

// DO_solver_workspace.hpp:4-44
struct DO_Workspace {
    int nInstances, total_size;
    std::vector<double> U, lambda_bar;  // [nInstances][total_size]
    DO_Workspace(int n, int total) : nInstances(n), total_size(total), U((size_t)n * total, 0.0), lambda_bar((size_t)n * total, 0.0) {}
};

struct Dividends {  // device_solver.hpp:409-413
    std::vector<double> dates, amounts, percentages;
};

// Put boundary data (hadi.h, enum hadi_option_type) -- NOT in the reference, whose only boundary class is call-specific
// (BoundaryConditions.hpp:7-12).  Passing the strikes as the optional last argument of a launcher below switches the batch
// to the put's boundary values; the caller supplies the put payoff in workspace.U / U_0 as usual.
using PutStrikes = std::vector<double>;

namespace detail {
inline hadi_problem make(int variant, int n, int m1, int m2, int N, double delta_t, double theta, double r_d, double r_f,
                         double rho, double sigma, double kappa, double eta, const GridViews &g, double *U,
                         const double *U_0, const Dividends *div, const PutStrikes *put = nullptr) {
    if (g.nInstances != n || g.m1 != m1 || g.m2 != m2) throw std::runtime_error("grid batch does not match (n, m1, m2)");
    hadi_problem p{};
    if (put) {
        if ((int)put->size() != n) throw std::runtime_error("put strikes do not match the batch");
        p.option_type = HADI_PUT;
        p.strike_i = put->data();
    }
    p.n_instances = n; p.m1 = m1; p.m2 = m2; p.variant = variant; p.memspace = HADI_MEM_HOST;
    p.N = N; p.delta_t = delta_t; p.theta = theta;
    p.r_d = r_d; p.r_f = r_f; p.rho = rho; p.sigma = sigma; p.kappa = kappa; p.eta = eta;
    p.vec_s = g.Vec_s.data(); p.vec_v = g.Vec_v.data(); p.delta_s = g.Delta_s.data(); p.delta_v = g.Delta_v.data();
    p.U = U; p.U_0 = U_0;
    if (div) {
        p.num_dividends = (int)div->dates.size();
        p.dividend_dates = div->dates.data(); p.dividend_amounts = div->amounts.data();
        p.dividend_percentages = div->percentages.data();
    }
    return p;
}
inline void check(const Handle &h, int rc) {
    if (rc) throw std::runtime_error(hadi_last_error(h.ctx));
}
}  // namespace detail

// device_solver.hpp:52-185 (S_0, V_0 are doubles here; the reference declares them int, :56-57)
inline void parallel_DO_solve(Handle &h, int nInstances, double S_0, double V_0, int m1, int m2, int N, double /*T*/,
                              double delta_t, double theta, double r_d, double r_f, double rho, double sigma, double kappa,
                              double eta, const GridViews &deviceGrids, DO_Workspace &workspace, std::vector<double> &base_prices,
                              const PutStrikes *put = nullptr) {
    base_prices.resize(nInstances);
    hadi_problem p = detail::make(HADI_EU, nInstances, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, deviceGrids,
                                  workspace.U.data(), nullptr, nullptr, put);
    detail::check(h, hadi_parallel_DO_solve(h.ctx, &p, S_0, V_0, base_prices.data()));
}

// jacobian_computation.cpp:368-448 and the three variants; workspace.U holds the initial condition (the reference's
// callers do deep_copy(workspace.U, U_0) first, heston_calibration.cpp:216)
inline void compute_base_prices(Handle &h, double S_0, double V_0, double /*T*/, double r_d, double r_f, double rho,
                                double sigma, double kappa, double eta, int m1, int m2, int /*total_size*/, int N, double theta,
                                double delta_t, int num_strikes, const GridViews &deviceGrids, DO_Workspace &workspace,
                                std::vector<double> &base_prices) {
    base_prices.resize(num_strikes);
    hadi_problem p = detail::make(HADI_EU, num_strikes, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, deviceGrids,
                                  workspace.U.data(), nullptr, nullptr);
    detail::check(h, hadi_compute_base_prices(h.ctx, &p, S_0, V_0, base_prices.data()));
}
inline void compute_base_prices_american(Handle &h, double S_0, double V_0, double, double r_d, double r_f, double rho,
                                         double sigma, double kappa, double eta, int m1, int m2, int, int N, double theta,
                                         double delta_t, int num_strikes, const GridViews &deviceGrids,
                                         const std::vector<double> &U_0, DO_Workspace &workspace,
                                         std::vector<double> &base_prices) {
    base_prices.resize(num_strikes);
    hadi_problem p = detail::make(HADI_AM, num_strikes, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, deviceGrids,
                                  workspace.U.data(), U_0.data(), nullptr);
    p.lambda_bar = workspace.lambda_bar.data();
    detail::check(h, hadi_compute_base_prices_american(h.ctx, &p, S_0, V_0, base_prices.data()));
}
inline void compute_base_prices_dividends(Handle &h, double S_0, double V_0, double, double r_d, double r_f, double rho,
                                          double sigma, double kappa, double eta, int m1, int m2, int, int N, double theta,
                                          double delta_t, int num_strikes, const GridViews &deviceGrids, DO_Workspace &workspace,
                                          const Dividends &div, std::vector<double> &base_prices) {
    base_prices.resize(num_strikes);
    hadi_problem p = detail::make(HADI_DIV, num_strikes, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, deviceGrids,
                                  workspace.U.data(), nullptr, &div);
    detail::check(h, hadi_compute_base_prices_dividends(h.ctx, &p, S_0, V_0, base_prices.data()));
}
inline void compute_base_prices_american_dividends(Handle &h, double S_0, double V_0, double, double r_d, double r_f,
                                                   double rho, double sigma, double kappa, double eta, int m1, int m2, int,
                                                   int N, double theta, double delta_t, int num_strikes,
                                                   const GridViews &deviceGrids, const std::vector<double> &U_0,
                                                   DO_Workspace &workspace, const Dividends &div,
                                                   std::vector<double> &base_prices, const PutStrikes *put = nullptr) {
    base_prices.resize(num_strikes);
    hadi_problem p = detail::make(HADI_AM_DIV, num_strikes, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta,
                                  deviceGrids, workspace.U.data(), U_0.data(), &div, put);
    p.lambda_bar = workspace.lambda_bar.data();
    detail::check(h, hadi_compute_base_prices_american_dividends(h.ctx, &p, S_0, V_0, base_prices.data()));
}

// jacobian_computation.cpp:204-364 and the three variants.  J: [num_strikes][5], columns kappa, eta, sigma, rho, v0.
inline void compute_jacobian(Handle &h, double S_0, double V_0, double, double r_d, double r_f, double rho, double sigma,
                             double kappa, double eta, int m1, int m2, int, int N, double theta, double delta_t,
                             int num_strikes, const GridViews &deviceGrids, const std::vector<double> &U_0,
                             std::vector<double> &J, std::vector<double> &base_prices, double eps = 1e-6,
                             int variant = HADI_EU, const Dividends *div = nullptr) {
    J.resize((size_t)num_strikes * 5);
    base_prices.resize(num_strikes);
    hadi_problem p = detail::make(variant, num_strikes, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, deviceGrids,
                                  nullptr, U_0.data(), div);
    int rc;
    switch (variant) {
        case HADI_AM: rc = hadi_compute_jacobian_american(h.ctx, &p, S_0, V_0, eps, J.data(), base_prices.data()); break;
        case HADI_DIV: rc = hadi_compute_jacobian_dividends(h.ctx, &p, S_0, V_0, eps, J.data(), base_prices.data()); break;
        case HADI_AM_DIV: rc = hadi_compute_jacobian_american_dividends(h.ctx, &p, S_0, V_0, eps, J.data(), base_prices.data()); break;
        default: rc = hadi_compute_jacobian(h.ctx, &p, S_0, V_0, eps, J.data(), base_prices.data()); break;
    }
    detail::check(h, rc);
}
inline void compute_jacobian_american(Handle &h, double S_0, double V_0, double T, double r_d, double r_f, double rho,
                                      double sigma, double kappa, double eta, int m1, int m2, int ts, int N, double theta,
                                      double delta_t, int num_strikes, const GridViews &g, const std::vector<double> &U_0,
                                      std::vector<double> &J, std::vector<double> &base_prices, double eps = 1e-6) {
    compute_jacobian(h, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, ts, N, theta, delta_t, num_strikes, g, U_0, J,
                     base_prices, eps, HADI_AM, nullptr);
}
inline void compute_jacobian_dividends(Handle &h, double S_0, double V_0, double T, double r_d, double r_f, double rho,
                                       double sigma, double kappa, double eta, int m1, int m2, int ts, int N, double theta,
                                       double delta_t, int num_strikes, const GridViews &g, const std::vector<double> &U_0,
                                       const Dividends &div, std::vector<double> &J, std::vector<double> &base_prices,
                                       double eps = 1e-6) {
    compute_jacobian(h, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, ts, N, theta, delta_t, num_strikes, g, U_0, J,
                     base_prices, eps, HADI_DIV, &div);
}
inline void compute_jacobian_american_dividends(Handle &h, double S_0, double V_0, double T, double r_d, double r_f,
                                                double rho, double sigma, double kappa, double eta, int m1, int m2, int ts,
                                                int N, double theta, double delta_t, int num_strikes, const GridViews &g,
                                                const std::vector<double> &U_0, const Dividends &div, std::vector<double> &J,
                                                std::vector<double> &base_prices, double eps = 1e-6) {
    compute_jacobian(h, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, ts, N, theta, delta_t, num_strikes, g, U_0, J,
                     base_prices, eps, HADI_AM_DIV, &div);
}

// jacobian_computation.cpp:107-195
inline void compute_parameter_update_on_device(const std::vector<double> &J, const std::vector<double> &residuals,
                                               double lambda, std::vector<double> &delta) {
    delta.resize(5);
    if (hadi_compute_parameter_update((int)residuals.size(), J.data(), residuals.data(), lambda, delta.data()))
        throw std::runtime_error("hadi_compute_parameter_update failed");
}

}  // namespace hadi_host
