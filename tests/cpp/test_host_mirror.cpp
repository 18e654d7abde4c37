// C++ parity test through include/hadi_host.hpp, written after the reference's own device test
// (test_deviceCallable_Do_solver, src/device_solver.cpp:598-813: 20 strikes 85..104 on the 50x25 grid, 20 steps) and its
// Jacobian test (test_jacobian_method, src/jacobian_computation.cpp:1513: 25x20 grid, strikes 90..94).
// Expected values: tests/golden/reference_known_answers.json.  Built and run by tests/test_cpp_host.py.
#include <algorithm>
#include <cmath>
#include <cstdio>

#include "hadi_host.hpp"

using namespace hadi_host;

static int fails = 0;
static void expect(const char *what, double got, double want, double tol) {
    const bool ok = std::fabs(got - want) <= tol;
    std::printf("%s %-46s got %.16g want %.16g\n", ok ? "ok  " : "FAIL", what, got, want);
    if (!ok) fails++;
}

int main() {
    const double S_0 = 100.0, V_0 = 0.04, T = 1.0, r_d = 0.025, r_f = 0.0, rho = -0.9, sigma = 0.3, kappa = 1.5, eta = 0.04;
    const double theta = 0.8;
    Handle h(0);
    {   // ---- batched European prices, then the American / dividend launchers on the same grids ----
        const int m1 = 50, m2 = 25, nInstances = 20, N = 20, total_size = (m1 + 1) * (m2 + 1);
        const double delta_t = T / N;
        std::vector<double> strikes(nInstances);
        for (int i = 0; i < nInstances; ++i) strikes[i] = 85.0 + i;
        GridViews grids;
        buildMultipleGridViews(grids, nInstances, m1, m2);
        std::vector<double> U_0((size_t)nInstances * total_size);
        for (int i = 0; i < nInstances; ++i) {
            const double K = strikes[i];
            Grid tempGrid(m1, 8 * K, S_0, K, K / 5, m2, 5.0, V_0, 5.0 / 500);
            grids.set(i, tempGrid);
            for (int j = 0; j <= m2; j++)
                for (int k = 0; k <= m1; k++) U_0[(size_t)i * total_size + k + j * (m1 + 1)] = std::max(tempGrid.Vec_s[k] - K, 0.0);
        }
        DO_Workspace workspace(nInstances, total_size);
        std::vector<double> base_prices;
        workspace.U = U_0;
        parallel_DO_solve(h, nInstances, S_0, V_0, m1, m2, N, T, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids, workspace,
                          base_prices);
        expect("parallel_DO_solve K=85", base_prices[0], 19.36766348353193, 1e-9);
        expect("parallel_DO_solve K=86", base_prices[1], 18.57361239232885, 1e-9);
        expect("parallel_DO_solve K=87", base_prices[2], 17.7901173119001, 1e-9);
        expect("parallel_DO_solve K=100", base_prices[15], 8.8512320311290900, 1e-9);
        workspace.U = U_0;
        compute_base_prices(h, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta, delta_t, nInstances,
                            grids, workspace, base_prices);
        expect("compute_base_prices K=100", base_prices[15], 8.8512320311290900, 1e-9);
        workspace.U = U_0;
        compute_base_prices_american(h, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta, delta_t,
                                     nInstances, grids, U_0, workspace, base_prices);
        expect("compute_base_prices_american K=100", base_prices[15], 8.8512494014258465, 1e-9);
        Dividends div{{0.2, 0.4, 0.6, 0.8}, {0.5, 0.3, 0.2, 0.1}, {0.02, 0.02, 0.02, 0.02}};
        workspace.U = U_0;
        compute_base_prices_dividends(h, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta, delta_t,
                                      nInstances, grids, workspace, div, base_prices);
        expect("compute_base_prices_dividends K=100", base_prices[15], 3.8509622259329976, 1e-9);
        workspace.U = U_0;
        compute_base_prices_american_dividends(h, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                                               delta_t, nInstances, grids, U_0, workspace, div, base_prices);
        expect("compute_base_prices_american_dividends K=100", base_prices[15], 5.4303035460631257, 1e-9);
        expect("compute_base_prices_american_dividends K=95", base_prices[10], 8.5105730742666701, 1e-9);
        // ---- put boundary data (extension, hadi.h): European put-call parity against the discrete discount factor of
        // the Douglas step, and the American put with dividends (BASELINE config 3's option type) above the European one
        std::vector<double> call = base_prices, P_0((size_t)nInstances * total_size);
        workspace.U = U_0;
        parallel_DO_solve(h, nInstances, S_0, V_0, m1, m2, N, T, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids, workspace, call);
        for (int i = 0; i < nInstances; ++i)
            for (int j = 0; j <= m2; j++)
                for (int k = 0; k <= m1; k++)
                    P_0[(size_t)i * total_size + k + j * (m1 + 1)] = std::max(strikes[i] - grids.Vec_s[(size_t)i * (m1 + 1) + k], 0.0);
        std::vector<double> put, amput;
        workspace.U = P_0;
        parallel_DO_solve(h, nInstances, S_0, V_0, m1, m2, N, T, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids, workspace, put,
                          &strikes);
        const double hr = 0.5 * r_d * theta * delta_t;
        const double g = ((1 - delta_t * r_d + hr) / (1 + hr) + hr) / (1 + hr);
        expect("put-call parity K=100 (discrete forward)", call[15] - put[15], S_0 - 100.0 * std::pow(g, N), 3e-6);
        workspace.U = P_0;
        compute_base_prices_american_dividends(h, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                                               delta_t, nInstances, grids, P_0, workspace, div, amput, &strikes);
        expect("American put with dividends >= payoff K=104", std::min(amput[19] - (104.0 - S_0), 0.0), 0.0, 0.0);
        expect("American put with dividends > European put K=100", (amput[15] > put[15]) ? 1.0 : 0.0, 1.0, 0.0);
    }
    {   // ---- Jacobian row of test_jacobian_method ----
        const int m1 = 25, m2 = 20, num_strikes = 5, N = 20, total_size = (m1 + 1) * (m2 + 1);
        const double delta_t = T / N;
        GridViews grids;
        buildMultipleGridViews(grids, num_strikes, m1, m2);
        std::vector<double> U_0((size_t)num_strikes * total_size);
        for (int i = 0; i < num_strikes; ++i) {
            const double K = 90.0 + i;
            Grid g(m1, 8 * K, S_0, K, K / 5, m2, 5.0, V_0, 5.0 / 500);
            grids.set(i, g);
            for (int j = 0; j <= m2; j++)
                for (int k = 0; k <= m1; k++) U_0[(size_t)i * total_size + k + j * (m1 + 1)] = std::max(g.Vec_s[k] - K, 0.0);
        }
        std::vector<double> J, base;
        compute_jacobian(h, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta, delta_t, num_strikes,
                         grids, U_0, J, base, 1e-6);
        expect("compute_jacobian base K=90", base[0], 15.47729535057304, 1e-9);
        const double want[5] = {-0.066643428553, 32.885965326912, 0.532589714553, -0.431705997173, 36.356131383641};
        const char *nm[5] = {"dP/dkappa", "dP/deta", "dP/dsigma", "dP/drho", "dP/dv0"};
        for (int q = 0; q < 5; q++) expect(nm[q], J[q], want[q], 2e-5);
        std::vector<double> resid(num_strikes, 0.01), delta;
        compute_parameter_update_on_device(J, resid, 0.01, delta);
        std::printf("LM update: %g %g %g %g %g\n", delta[0], delta[1], delta[2], delta[3], delta[4]);
    }
    std::printf(fails ? "FAILED (%d)\n" : "all C++ host-mirror checks passed\n", fails);
    return fails ? 1 : 0;
}
