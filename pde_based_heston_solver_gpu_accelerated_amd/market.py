"""Market-data helpers used by the reference's calibration drivers: the `BlackScholes` class of src/bs.hpp
(call price, vega, implied volatility by Newton with bisection fallback) and the synthetic market
generators `generate_market_data{,_with_dividends}` (src/bs.hpp:44-118).  Pure host code."""
import math

import numpy as np

MARKET_VOL = 0.2  # "Fixed market volatility for synthetic data generation", bs.hpp:52


def call_price(S, K, r, v, T):
    """bs.hpp:36-43 (erfc form; the CP flag of the reference is unused there as well)."""
    sqrt_T = math.sqrt(T)
    vol_sqrt_T = v * sqrt_T
    d1 = (math.log(S / K) + (r + 0.5 * v * v) * T) / vol_sqrt_T
    d2 = d1 - vol_sqrt_T
    return S * math.erfc(-d1 / math.sqrt(2.0)) / 2.0 - K * math.exp(-r * T) * math.erfc(-d2 / math.sqrt(2.0)) / 2.0


def call_vega(S, K, r, v, T, CP=1):
    """bs.hpp:122-125."""
    d = (math.log(S / K) + (r + 0.5 * v * v) * T) / (v * math.sqrt(T))
    return CP * S * math.exp(-d * d / 2.0) * math.sqrt(T / (2.0 * math.pi))


def reverse_BS_dic(S, K, r, T, C_target, epsilon, a, b, max_iter=1000):
    """Implied volatility by bisection, bs.hpp:127-149."""
    x = (b + a) / 2
    C = call_price(S, K, r, x, T)
    it = 0
    while abs(C - C_target) > epsilon and it < max_iter:
        C = call_price(S, K, r, x, T)
        if C > C_target:
            b = x
        else:
            a = x
        x = (b + a) / 2
        it += 1
    return x


def reverse_BS(S, K, r, T, v_0, C_target, epsilon, max_newton=200):
    """Implied volatility: Newton on vega, falling back to bisection on [0.001, 1] when vega vanishes
    (bs.hpp:151-175; the reference's Newton loop is unbounded, here it is capped and falls back too)."""
    x = v_0
    C = call_price(S, K, r, x, T)
    fail, it = False, 0
    while abs(C - C_target) > epsilon:
        C = call_price(S, K, r, x, T)
        V = call_vega(S, K, r, x, T)
        it += 1
        if abs(V) < 1e-10 or it > max_newton or not (x > 0):
            fail = True
            break
        x -= (C - C_target) / V
    if fail:
        x = reverse_BS_dic(S, K, r, T, C_target, epsilon, 0.001, 1.0)
    return x


def generate_market_data(S_0, T, r_d, strikes):
    """Synthetic market: Black-Scholes calls at 20 % volatility (bs.hpp:44-64)."""
    return np.array([call_price(S_0, K, r_d, MARKET_VOL, T) for K in strikes])


def generate_market_data_with_dividends(S_0, T, r_d, strikes, dividend_dates, dividend_amounts, dividend_percentages):
    """Same on the dividend-adjusted spot (escrowed-dividend model, bs.hpp:66-118)."""
    S_adj = S_0
    for date, amount, pct in zip(dividend_dates, dividend_amounts, dividend_percentages):
        if date < T:
            S_adj -= amount * math.exp(-r_d * date)
            S_adj -= (S_0 * pct) * math.exp(-r_d * date)
    return np.array([call_price(S_adj, K, r_d, MARKET_VOL, T) for K in strikes])
