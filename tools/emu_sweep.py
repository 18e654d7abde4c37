#!/usr/bin/env python3
"""CPU-only campaign (no GPU): the random problems of tests/test_emu_kernel_logic.py::test_random_shapes_and_kernel_choices_vs_oracle
over many more seeds -- the product's kernel source under the wave emulator against the oracle's full field.
    python tools/emu_sweep.py [first_seed] [seeds]       (30 cases per seed, ~8 s per seed)
Builds nothing: run the CPU suite once first (it compiles tests/emu/libhadi_emu.so)."""
import os, sys, random, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_emu_kernel_logic as T
import common as Cm
emu = C.CDLL(T.EMU_SO)
bad = 0; total = 0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for seed in range(first, first + (int(sys.argv[2]) if len(sys.argv) > 2 else 40)):
    rng = random.Random(5000 + seed)
    for k in range(30):
        c = T._random_case(rng)
        for _ in range(50):
            d = np.diff(Cm.oracle_grids(c["m1"], 8, c["strikes"])[0], axis=1)
            if np.maximum(d[:, 1:] / d[:, :-1], d[:, :-1] / d[:, 1:]).max() <= 30.0: break
            c["strikes"] = [rng.uniform(85, 115) for _ in c["strikes"]]
        for key, val in c["tuning"].items(): emu.emu_set_tuning(key.encode(), val)
        try:
            T._run(emu, c["m1"], c["m2"], c["N"], c["strikes"], c["variant"], c["target_waves"], r_f=c["r_f"], scheme=c["scheme"], put=c["put"], small=c["small"], tol=1e-10)
        except AssertionError as e:
            bad += 1; print("BAD seed", seed, "case", k, c, str(e)[:200], flush=True)
        finally:
            emu.emu_set_tuning(b"reset", 0)
        total += 1
    print("seed", seed, "done", total, "cases", bad, "bad", flush=True)
