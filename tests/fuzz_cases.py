"""Deterministic random parity cases shared by tools/fuzz_parity.py (the campaign) and the regression tests that pin the
cases a campaign flagged (tests/test_gpu_regressions.py): case `index` of seed `seed` is always the same problem.

The generator consumes its random stream in a FIXED order -- do not reorder the draws: the recorded (seed, index) pairs
of past campaigns (gpurun_out/fuzz_*.log) would point at different problems."""
import random

EU, AM, DIV, AM_DIV = 0, 1, 2, 3
NAMES = {EU: "EU", AM: "AM", DIV: "DIV", AM_DIV: "AM_DIV"}


def draw_case(rng, small=False, wide=False):
    """One case as a dict: grid shape, time steps, batch size, variant, parameters, strikes, call/put, state precision and
    the kernel-selection overrides the campaign applied (tuning: key -> value)."""
    m1 = rng.choice([rng.randint(20, 64), rng.randint(65, 128), rng.randint(129, 256), rng.randint(257, 512), rng.randint(513, 1024)])
    if small: m1 = rng.randint(8, 128)  # LDS-resident shapes only
    if wide: m1 = rng.choice([rng.randint(513, 1024), 1024, 513])  # two wavefronts per row only
    m2 = rng.randint(8, 300) if rng.random() < 0.2 else rng.randint(8, min(m1, 300))  # (m2 > m1 now and then)
    if small: m2 = rng.randint(4, 32)
    N = rng.randint(2, 12)
    n = rng.choice([1, 2, 3, 5, 9, 40, 130, 300]) if m1 * m2 < 40000 else rng.choice([1, 2, 3, 5, 9, 70])
    variant = rng.choice([EU, AM, DIV, AM_DIV])
    r_f = rng.choice([0.0, 0.01, 0.03])
    model = (rng.uniform(-0.95, 0.5), rng.uniform(0.1, 0.8), rng.uniform(0.3, 4.0), rng.uniform(0.01, 0.2))  # rho sigma kappa eta
    strikes = [rng.uniform(80, 120) for _ in range(n)]
    put = rng.random() < 0.35
    f32 = variant in (EU, DIV) and rng.random() < 0.2
    tuning = {}
    if rng.random() < 0.3: tuning["american_p"] = 0
    if rng.random() < 0.5: tuning["small_seq"] = 1  # (LDS-resident grids, European / dividends: the one-wavefront kernel)
    if rng.random() < (0.7 if wide else 0.3): tuning["strip"] = 1
    return dict(m1=m1, m2=m2, N=N, n=n, variant=variant, name=NAMES[variant], r_f=r_f, model=model, strikes=strikes, put=put,
                f32=f32, tuning=tuning)


def cases(seed, count, small=False, wide=False):
    rng = random.Random(seed)
    for index in range(count):
        c = draw_case(rng, small, wide)
        c["seed"], c["index"] = seed, index
        yield c


def case(seed, index, small=False, wide=False):
    for c in cases(seed, index + 1, small, wide):
        pass
    return c


def summary(c):
    return "%s%s%s m1=%d m2=%d N=%d n=%d r_f=%.2f" % (c["name"], " put" if c["put"] else "", " f32" if c["f32"] else "", c["m1"], c["m2"],
                                                       c["N"], c["n"], c["r_f"])
