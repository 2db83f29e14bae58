"""CPU-side checks of the boundary: both shared libraries load and export every symbol that
include/*.h declares (no compute without a GPU), the host mirror validates like
src/problem.rs:372-429, parses MPS like src/parse_mps.rs:565-643, and builds bit-identical
phase-1 inputs to the oracle's independent restatement of the setup."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ellp_amd
from ellp_amd import Bound, EllPError, MpsParsingError, Problem, _engine, parse_mps
from helpers import GOLDEN, known_answers, read_mps
from oracle import ellp_oracle as eo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KA = known_answers()


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ellp_[a-z0-9_]+)\s*\(", text)))


def test_engine_library_exports_every_declared_symbol():
    lib = _engine.lib()
    names = _declared_functions("ellp_hip.h")
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"libellp_hip.so does not export {n}"
    assert lib.ellp_hip_abi_version() == 1


def test_host_library_exports_every_declared_symbol():
    lib = ellp_amd.host_lib()
    for n in _declared_functions("ellp_host.h"):
        assert hasattr(lib, n), f"libellp_host.so does not export {n}"


def test_flag_constants_are_the_headers():
    """ellp_opts.flags: the Python constants are the #defines of include/ellp_hip.h, and every bit is documented at the field"""
    text = open(os.path.join(ROOT, "include", "ellp_hip.h")).read()
    defs = {n: int(v) for n, v in re.findall(r"#define\s+ELLP_FLAG_([A-Z_]+)\s+(\d+)", text)}
    assert defs == {"DENSE_PRICING": _engine.FLAG_DENSE_PRICING, "DUAL_MAX_VIOLATION": _engine.FLAG_DUAL_MAX_VIOLATION,
                    "PRIMAL_STEEPEST_EDGE": _engine.FLAG_PRIMAL_STEEPEST_EDGE, "NO_CERTIFY": _engine.FLAG_NO_CERTIFY,
                    "DUAL_BOUND_FLIPPING": _engine.FLAG_DUAL_BOUND_FLIPPING}
    assert sorted(defs.values()) == [1, 2, 4, 8, 16]
    for n in defs:
        assert text.count("ELLP_FLAG_" + n) >= 2, n  # the #define and the field's description


def test_default_opts_reproduce_reference_defaults():
    o = _engine.default_opts()
    assert o.max_iter == 1000  # primal…:21, dual…:22
    assert o.eps == 1e-10      # util.rs:1


def test_engine_argument_errors_without_gpu():
    """Argument validation happens before any device work (primal…:124-140)."""
    E = _engine
    A = np.eye(2).reshape(-1)
    fp = E.FlatProblem(2, 4, 4, np.concatenate([A, A]), np.zeros(4), np.zeros(2), np.ones(4, np.uint8),
                       np.zeros(4), np.zeros(4), np.zeros(4), [0], [1, 2, 3], [0, 0, 0])
    st, _, msg = E.primal_solve_with_initial(fp)
    assert st == E.ERR_BAD_DIMS and "invalid B, has 1 elements but 2 expected" in msg
    fp = E.FlatProblem(2, 4, 4, np.concatenate([A, A]), np.zeros(4), np.zeros(2), np.ones(4, np.uint8),
                       np.zeros(4), np.zeros(4), np.zeros(4), [0, 1], [2], [0])
    st, _, msg = E.primal_solve_with_initial(fp)
    assert st == E.ERR_BAD_DIMS and "invalid N, has 1 elements but 2 expected" in msg


# ---- src/problem.rs:372-429
def test_add_var():
    p = Problem()
    assert p.add_var(1.0, Bound.Free, "x") == 0
    assert p.num_vars == 1


def test_add_var_bad_bounds():
    with pytest.raises(EllPError):
        Problem().add_var(1.0, Bound.TwoSided(1.0, 0.0), "x")
    with pytest.raises(EllPError):
        Problem().add_var(1.0, Bound.Lower(float("inf")), "x")


def test_add_constraint_and_invalid_var():
    p = Problem()
    v = p.add_var(1.0, Bound.Free)
    p.add_constraint([(v, 1.0)], "Lte", 0.0)
    with pytest.raises(EllPError):
        Problem().add_constraint([(0, 1.0)], "Lte", 0.0)


def test_nonunique_var_names():
    p = Problem()
    p.add_var(0.0, Bound.Free, "x")
    with pytest.raises(EllPError):
        p.add_var(0.0, Bound.Free, "x")


def test_is_feasible():
    p = Problem()
    a = p.add_var(1.0, Bound.Lower(0.0))
    b = p.add_var(1.0, Bound.TwoSided(-1.0, 1.0))
    p.add_constraint([(a, 1.0), (b, 1.0)], "Lte", 2.0)
    assert p.is_feasible([1.0, 1.0])
    assert not p.is_feasible([-1.0, 1.0])
    assert not p.is_feasible([2.0, 1.0])


# ---- src/parse_mps.rs:565-643
def test_parse_mps_example():
    text = open(os.path.join(GOLDEN, "testprob.mps")).read()
    p = parse_mps(text)
    assert p.num_vars == 3 and p.num_constraints == 3
    flat = p._debug_phase1("primal")
    # XONE Upper(4), YTWO TwoSided(-1, 1), ZTHREE Lower(0) (file order)
    assert list(flat["kind"][:3]) == [2, 3, 1]
    assert flat["ub"][0] == 4.0 and (flat["lb"][1], flat["ub"][1]) == (-1.0, 1.0) and flat["lb"][2] == 0.0
    assert sorted(flat["b"].tolist()) == [5.0, 7.0, 10.0]


def test_parse_mps_errors():
    with pytest.raises(MpsParsingError):
        parse_mps("ROWS\n")
    with pytest.raises(MpsParsingError):
        parse_mps("NAME X\nROWS\n Q R1\nCOLUMNS\nRHS\nENDATA\n")


def _all_fixture_problems():
    out = [(p["name"], p) for p in KA["problems"]]
    out += [(n["name"], read_mps(os.path.join(GOLDEN, n["file"]))) for n in KA["netlib"]]
    return out


@pytest.mark.parametrize("solver", ["primal", "dual"])
@pytest.mark.parametrize("name,fx", _all_fixture_problems(), ids=[n for n, _ in _all_fixture_problems()])
def test_host_setup_matches_oracle_setup(name, fx, solver):
    """Two independent restatements (C++ host mirror, C oracle) of standard_form.rs +
    *_problem.rs must hand the loops the very same arrays."""
    hp = Problem.from_fixture(fx)._debug_phase1(solver)
    op = eo.Problem.from_fixture(fx)
    ph, err = eo.primal_phase1(op) if solver == "primal" else eo.dual_phase1(op)
    if ph is None:
        assert hp is None
        return
    v = ph.view()
    assert (hp["m"], hp["n"], hp["n_c"]) == (v.m, v.n, v.n_c)
    for k in ("A", "c", "b", "kind", "lb", "ub", "x", "B"):
        np.testing.assert_array_equal(hp[k], getattr(v, k), err_msg=k)
    np.testing.assert_array_equal(hp["N"], v.N[:v.nN])
    np.testing.assert_array_equal(hp["Nb"], v.Nb[:v.nN])
    if solver == "dual" and v.m > 0:
        np.testing.assert_array_equal(hp["y"], v.y)
        np.testing.assert_array_equal(hp["d"], v.d)


def test_mps_parser_matches_test_loader():
    """C++ parse_mps (product) and the test-side loader agree on the netlib fixtures."""
    for n in KA["netlib"]:
        path = os.path.join(GOLDEN, n["file"])
        a = parse_mps(open(path).read())._debug_phase1("primal")
        b = Problem.from_fixture(read_mps(path))._debug_phase1("primal")
        for k in ("A", "c", "b", "kind", "x", "B", "N"):
            np.testing.assert_array_equal(a[k], b[k])
