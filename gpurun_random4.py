import sys, os, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
import numpy as np
from oracle import ellp_oracle as eo
import test_gpu_random as T

def wide_fixture(rng):
    """feasible, bounded, mixed bound kinds, few rows and thousands of columns (several columns per pricing block)"""
    m = int(rng.integers(30, 80)); n = int(rng.integers(1200, 3200))
    integer = rng.random() < 0.4
    vars_, x0 = [], []
    for j in range(n):
        c = float(rng.integers(-4, 5)) if integer else float(rng.normal())
        lo = float(rng.integers(-3, 3)) if integer else float(rng.normal())
        w = float(rng.integers(1, 5)) if integer else float(abs(rng.normal()) + 0.1)
        u = rng.random()
        if u < 0.05:
            vars_.append([c, ["Fixed", lo, lo]]); x0.append(lo)
        elif u < 0.45:
            vars_.append([c, ["TwoSided", lo, lo + w]]); x0.append(lo + (float(rng.integers(0, int(w) + 1)) if integer else float(rng.random() * w)))
        elif u < 0.75:
            vars_.append([abs(c), ["Lower", lo, 0.0]]); x0.append(lo + (float(rng.integers(0, 3)) if integer else float(abs(rng.normal()))))
        elif u < 2.0:
            vars_.append([-abs(c), ["Upper", 0.0, lo]]); x0.append(lo - (float(rng.integers(0, 3)) if integer else float(abs(rng.normal()))))
        else:
            vars_.append([0.0, ["Free", 0.0, 0.0]]); x0.append(float(rng.normal()))
    cons = []
    dens = rng.choice([0.05, 0.3, 1.0])
    for i in range(m):
        a = np.where(rng.random(n) < dens, rng.integers(-3, 4, size=n).astype(float) if integer else rng.normal(size=n), 0.0)
        ax = float(np.dot(a, x0))
        op = str(rng.choice(["Lte", "Gte", "Eq"], p=[0.45, 0.4, 0.15]))
        slack = float(rng.integers(0, 4)) if integer else float(abs(rng.normal()))
        rhs = ax + slack if op == "Lte" else (ax - slack if op == "Gte" else ax)
        cons.append([[[j, float(a[j])] for j in range(n) if a[j] != 0.0], op, rhs])
    return {"vars": vars_, "constraints": cons}

t0 = time.time()
n, bad = T._campaign(int(sys.argv[1]), int(sys.argv[2]), wide_fixture)
print(n, "mismatches", len(bad), "time", time.time() - t0)
for b in bad[:40]: print(b)
