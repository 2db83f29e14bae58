"""tools/small_stamps.py [name] — where an iteration of the persistent small-LP kernel spends its time: solves a netlib
fixture through the user API with ELLP_SMALL_STAMPS=1 (k_small then sums 100 MHz ticks per phase; printed to stderr
when each engine is destroyed) for each workgroup size that has a thread per row (ELLP_SMALL_NT)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "adlittle"
code = ("import sys; sys.path.insert(0, %r)\n"
        "from ellp_amd import PrimalSimplexSolver, parse_mps\n"
        "r = PrimalSimplexSolver.new(None).solve(parse_mps(open(%r).read()))\n"
        "print(r.kind, r.iters)\n") % (root, os.path.join(root, "tests", "golden", "netlib", name + ".mps"))
for nt in ("64", "128", "256"):
    env = dict(os.environ, ELLP_SMALL_STAMPS="1", ELLP_SMALL_NT=nt)
    print("== ELLP_SMALL_NT=" + nt, flush=True)
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
