"""Shared test helpers: fixtures, a small MPS reader (data loader for the netlib fixtures),
result checks written the way the reference's test macros are (tests/problems/mod.rs:9-71)."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)


def read_mps(path):
    """Free-format MPS subset the reference accepts (src/parse_mps.rs:23-546): NAME, ROWS,
    COLUMNS (one entry per line), RHS, BOUNDS (UP/LO/FR), ENDATA.  Variables and rows are kept
    in FILE order (the reference iterates HashMaps, so its order is random per process and its
    tests pin the objective only).  Returns a fixture dict like known_answers.json's."""
    rows, row_order, cols, col_order = {}, [], {}, []
    section = None
    with open(path) as f:
        for raw in f:
            if not raw.strip():
                continue
            tok = raw.split()
            if raw[0] not in " \t":
                section = tok[0]
                if section == "ENDATA":
                    break
                continue
            if section == "ROWS":
                kind, name = tok
                rows[name] = {"kind": kind, "coeffs": {}, "rhs": None}
                row_order.append(name)
            elif section == "COLUMNS":
                var, row, val = tok
                if var not in cols:
                    cols[var] = {"obj": 0.0, "bound": None}
                    col_order.append(var)
                if rows[row]["kind"] == "N":
                    cols[var]["obj"] = float(val)
                else:
                    assert var not in rows[row]["coeffs"]
                    rows[row]["coeffs"][var] = float(val)
            elif section == "RHS":
                if len(tok) == 3:
                    tok = tok[1:]
                row, val = tok
                assert rows[row]["kind"] != "N" and rows[row]["rhs"] is None
                rows[row]["rhs"] = float(val)
            elif section == "BOUNDS":
                bt, col = tok[0], tok[2]
                val = float(tok[3]) if len(tok) > 3 else None
                cur = cols[col]["bound"]
                if bt == "UP":
                    new = ["Upper", 0.0, val]
                elif bt == "LO":
                    new = ["Lower", val, 0.0]
                elif bt == "FR" and val is None:
                    new = ["Free", 0.0, 0.0]
                else:
                    raise ValueError("invalid bound specification")
                if cur is None:
                    cols[col]["bound"] = new
                elif cur[0] == "Upper" and new[0] == "Lower":
                    cols[col]["bound"] = ["TwoSided", new[1], cur[2]]
                elif cur[0] == "Lower" and new[0] == "Upper":
                    cols[col]["bound"] = ["TwoSided", cur[1], new[2]]
                else:
                    raise ValueError("invalid bounds")
    index = {v: i for i, v in enumerate(col_order)}
    opmap = {"L": "Lte", "G": "Gte", "E": "Eq"}
    fx = {"vars": [], "constraints": []}
    for v in col_order:
        fx["vars"].append([cols[v]["obj"], cols[v]["bound"] or ["Lower", 0.0, 0.0]])
    for r in row_order:
        if rows[r]["kind"] == "N":
            continue
        coeffs = [[index[v], c] for v, c in rows[r]["coeffs"].items()]
        fx["constraints"].append([coeffs, opmap[rows[r]["kind"]], rows[r]["rhs"] or 0.0])
    return fx


def check_result(fx, status_name, obj, x, abs_eps=1e-8, rel_eps=1e-6):
    """assert_optimal! / assert_optimal_obj! / assert_unbounded! / assert_infeasible!"""
    chk = fx["check"]
    if chk == "infeasible":
        assert status_name == "infeasible", f"not infeasible: {status_name}"
    elif chk == "unbounded":
        assert status_name == "unbounded", f"not unbounded: {status_name}"
    elif chk == "optimal":
        assert status_name == "optimal", f"not optimal: {status_name}"
        assert abs(obj - fx["obj"]) < abs_eps, f"obj: {obj}, expected: {fx['obj']}"
        assert len(x) == len(fx["x"])
        for a, b in zip(x, fx["x"]):
            assert abs(a - b) < abs_eps, f"x_i: {a}, expected: {b}"
    elif chk == "optimal_obj":
        assert status_name == "optimal", f"not optimal: {status_name}"
        e = fx["obj"]
        assert abs(obj - e) < abs_eps or abs(obj / e - 1.0) < rel_eps, f"obj: {obj}, expected: {e}"
    else:
        raise AssertionError(chk)


def collect_results(q, procs, world, timeout):
    """One result per rank from the queue of a spawned world; gives up as soon as a rank has died without
    reporting (a crashed child must not cost the whole timeout — on the GPU box that is minutes of budget)."""
    import queue as _queue
    import time as _time
    results, deadline = [], _time.monotonic() + timeout
    while len(results) < world:
        try:
            results.append(q.get(timeout=1.0))
            continue
        except _queue.Empty:
            pass
        if _time.monotonic() > deadline:
            raise TimeoutError(f"{len(results)} of {world} ranks reported within {timeout} s")
        dead = [p.exitcode for p in procs if not p.is_alive() and p.exitcode not in (0, None)]
        if dead:
            _time.sleep(1.0)  # what the survivors still had in flight
            while len(results) < world:
                try:
                    results.append(q.get(timeout=0.2))
                except _queue.Empty:
                    break
            if len(results) < world:
                raise RuntimeError(f"a rank died (exit codes {dead}) before reporting; {len(results)} of {world} results")
    return results


def blockdiag(fx, copies):
    """`copies` independent copies of an LP as ONE problem (block-diagonal constraints, summed objective): its
    optimum is copies x the LP's, its bases are block-diagonal arrangements of the LP's own (ill-conditioned)
    bases — the way to get real, sparse, degenerate LPs with more than 128 rows out of the reference's three
    netlib fixtures (tests/problems/mod.rs:657-674), at a size the LU-per-iteration oracle still solves."""
    n = len(fx["vars"])
    out = {"vars": [], "constraints": []}
    for k in range(copies):
        out["vars"] += [[obj, list(bound)] for obj, bound in fx["vars"]]
        out["constraints"] += [[[[j + k * n, a] for j, a in coeffs], op, rhs] for coeffs, op, rhs in fx["constraints"]]
    return out


def permuted_fixture(fx, rng):
    """The same LP with its variables and constraints in another order (the reference builds its problems by
    iterating HashMaps, tests/problems/mod.rs:657-674, so every order occurs)."""
    import numpy as np
    n = len(fx["vars"])
    perm = rng.permutation(n)            # new index k holds old variable perm[k]
    inv = np.empty(n, dtype=int)
    inv[perm] = np.arange(n)
    rows = [fx["constraints"][i] for i in rng.permutation(len(fx["constraints"]))]
    return {"vars": [fx["vars"][j] for j in perm],
            "constraints": [[[[int(inv[j]), a] for j, a in coeffs], op, rhs] for coeffs, op, rhs in rows]}


def fixture_violation(fx, x):
    """Largest violation of a fixture's constraints (relative to 1 + |rhs|) and of its variable bounds (absolute) at the
    point x — what Problem::is_feasible (src/problem.rs:108-154, :237-250) tests with the absolute EPS = 1e-10, as a
    number, so that a test can state its own tolerance."""
    worst_c, worst_b = 0.0, 0.0
    for coeffs, op, rhs in fx["constraints"]:
        lhs = sum(a * x[j] for j, a in coeffs)
        if op == "Lte":
            v = lhs - rhs
        elif op == "Gte":
            v = rhs - lhs
        else:
            v = abs(lhs - rhs)
        worst_c = max(worst_c, v / (1.0 + abs(rhs)))
    for j, (obj, (kind, lb, ub)) in enumerate(fx["vars"]):
        if kind in ("Lower", "TwoSided"):
            worst_b = max(worst_b, lb - x[j])
        if kind in ("Upper", "TwoSided"):
            worst_b = max(worst_b, x[j] - ub)
        if kind == "Fixed":
            worst_b = max(worst_b, abs(x[j] - lb))
    return worst_c, worst_b

