// solvers.cpp — PrimalSimplexSolver / DualSimplexSolver drivers.
//   solve():                primal_simplex_solver.rs:32-93, dual_simplex_solver.rs:33-108 (host)
//   solve_with_initial():   the seam.  Checks that stay on the host are the ones the reference
//                           does before its loop (m == 0 -> trivial solver); everything else is
//                           one call through the C ABI into the HIP engine.  No CPU loop exists
//                           here: if the engine cannot run, the error is raised.
#include <cmath>

#include "ellp.h"

#include <cstdlib>

namespace ellp {

namespace {

struct Flat {
    std::vector<std::uint8_t> kind;
    std::vector<double> lb, ub;
    std::vector<std::int64_t> B, N;
    std::vector<std::uint8_t> Nb;
};

Flat flatten(const StandardForm &sf, const Point &pt) {
    Flat f;
    const size_t nc = sf.bounds.size();
    f.kind.resize(nc);
    f.lb.resize(nc);
    f.ub.resize(nc);
    for (size_t i = 0; i < nc; ++i) {
        f.kind[i] = static_cast<std::uint8_t>(sf.bounds[i].kind);
        f.lb[i] = sf.bounds[i].lb;
        f.ub[i] = (sf.bounds[i].kind == Bound::Fixed) ? sf.bounds[i].lb : sf.bounds[i].ub;
    }
    for (const auto &b : pt.B) f.B.push_back(static_cast<std::int64_t>(b.index));
    for (const auto &n : pt.N) {
        f.N.push_back(static_cast<std::int64_t>(n.index));
        f.Nb.push_back(static_cast<std::uint8_t>(n.bound));
    }
    return f;
}

void unflatten(const Flat &f, Point &pt) {
    for (size_t i = 0; i < pt.B.size(); ++i) pt.B[i].index = static_cast<size_t>(f.B[i]);
    for (size_t j = 0; j < pt.N.size(); ++j) {
        pt.N[j].index = static_cast<size_t>(f.N[j]);
        pt.N[j].bound = static_cast<NonbasicBound>(f.Nb[j]);
    }
}

ellp_opts make_opts(std::uint64_t max_iter, const EngineOptions &e) {
    ellp_opts o;
    ellp_default_opts(&o);
    o.max_iter = max_iter;
    o.eps = EPS;
    o.device = e.device;
    o.refactor_period = e.refactor_period;
    o.btran_mode = e.btran_mode;
    o.poll_interval = e.poll_interval;
    o.pipeline = e.pipeline;
    o.partial_segments = e.partial_segments;
    o.flags = e.flags;
    return o;
}

SolutionStatus to_status(ellp_status s, const char *err) {
    switch (s) {
    case ELLP_OPTIMAL: return SolutionStatus::Optimal;
    case ELLP_INFEASIBLE: return SolutionStatus::Infeasible;
    case ELLP_UNBOUNDED: return SolutionStatus::Unbounded;
    case ELLP_MAXITER: return SolutionStatus::MaxIter;
    case ELLP_ERR_BAD_DIMS:
    case ELLP_ERR_SINGULAR: throw EllPError(err);  // Err(EllPError), primal…:124-140, :175-179
    case ELLP_ERR_DEVICE: throw std::runtime_error(std::string("HIP engine unavailable: ") + err);
    default: throw EllPPanic(err);
    }
}

}  // namespace

// primal_simplex_solver.rs:95-236
SolutionStatus PrimalSimplexSolver::solve_with_initial(const StandardForm &sf, Point &pt, std::uint64_t *iters) const {
    if (iters) *iters = 0;
    if (sf.rows() == 0) {  // :118-122
        if (!pt.B.empty()) throw EllPPanic("assertion failed: B.is_empty()");
        return solve_trivial_problem(sf, pt.x, pt.N, true);
    }
    Flat f = flatten(sf, pt);
    const ellp_opts o = make_opts(max_iter_, engine_);
    ellp_stats st{};
    char err[512] = {0};
    const ellp_status s = ellp_primal_solve_with_initial(
        static_cast<std::int64_t>(sf.rows()), static_cast<std::int64_t>(sf.cols()),
        static_cast<std::int64_t>(sf.bounds.size()), sf.A.a.data(), sf.c.data(), sf.b.data(), f.kind.data(),
        f.lb.data(), f.ub.data(), pt.x.data(), f.B.data(), static_cast<std::int64_t>(f.B.size()), f.N.data(),
        f.Nb.data(), static_cast<std::int64_t>(f.N.size()), &o, &st, err, sizeof(err));
    if (iters) *iters = st.iters;
    const SolutionStatus out = to_status(s, err);
    unflatten(f, pt);
    return out;
}

namespace {

// owns a resident engine for the two phases of one primal solve
struct EngineHandle {
    ellp_engine *e = nullptr;
    ~EngineHandle() { if (e) ellp_engine_destroy(e); }
};

// one phase on a resident engine: run, then bring the point back (x, B, N as at the seam)
SolutionStatus run_resident(ellp_engine *e, std::uint64_t max_iter, Flat &f, Point &pt, std::uint64_t *iters) {
    ellp_stats st{};
    char err[512] = {0};
    const ellp_status s = ellp_engine_run(e, max_iter, &st, err, sizeof(err));
    if (iters) *iters = st.iters;
    SolutionStatus out = to_status(s, err);
    const ellp_status rs = ellp_engine_read_point(e, pt.x.data(), f.B.data(), f.N.data(), f.Nb.data(), nullptr,
                                                  nullptr, err, sizeof(err));
    if (rs != ELLP_OPTIMAL) out = to_status(rs, err);  // an error, or the status the still open last iteration ended in
    unflatten(f, pt);
    return out;
}

}  // namespace

// primal_simplex_solver.rs:32-93.  The two solve_with_initial calls of the reference become two
// slices of ONE resident engine: after phase 1 only the costs and bounds are replaced on the
// device (ellp_engine_rephase, primal_problem.rs:263-291) — the matrix, the basis and B^-1 stay in
// HBM.  Problems the seam never sends to the device (m == 0, no nonbasic column) take the plain path.
SolverResult PrimalSimplexSolver::solve(Problem prob) const {
    SolverResult res;
    auto p1 = PrimalPhase1::from_problem(std::move(prob));
    if (!p1) { res.kind = SolverResult::Infeasible; return res; }
    PrimalPhase1 phase_1 = std::move(*p1);
    const bool resident = phase_1.std_form.rows() > 0 && !phase_1.point.N.empty();
    EngineHandle eng;
    Flat f1;
    SolutionStatus s1;
    if (resident) {
        f1 = flatten(phase_1.std_form, phase_1.point);
        const StandardForm &sf = phase_1.std_form;
        const ellp_opts o = make_opts(max_iter_, engine_);
        char err[512] = {0};
        const ellp_status cs = ellp_engine_create(
            ELLP_ENGINE_PRIMAL, static_cast<std::int64_t>(sf.rows()), static_cast<std::int64_t>(sf.cols()),
            static_cast<std::int64_t>(sf.bounds.size()), sf.A.a.data(), sf.c.data(), sf.b.data(), f1.kind.data(),
            f1.lb.data(), f1.ub.data(), phase_1.point.x.data(), f1.B.data(), static_cast<std::int64_t>(f1.B.size()),
            f1.N.data(), f1.Nb.data(), static_cast<std::int64_t>(f1.N.size()), nullptr, nullptr, &o, &eng.e, err,
            sizeof(err));
        if (cs != ELLP_OPTIMAL) to_status(cs, err);  // throws: Err(EllPError) / panic / device
        s1 = run_resident(eng.e, max_iter_, f1, phase_1.point, &res.iters_phase1);
    } else {
        s1 = solve_with_initial(phase_1.std_form, phase_1.point, &res.iters_phase1);
    }
    switch (s1) {
    case SolutionStatus::Optimal: {
        const double obj = phase_1.obj();
        if (!(obj > -EPS)) throw EllPPanic("assertion failed: obj > -EPS");
        if (!(obj < EPS)) { res.kind = SolverResult::Infeasible; return res; }
        break;
    }
    case SolutionStatus::Infeasible: res.kind = SolverResult::Infeasible; return res;
    case SolutionStatus::Unbounded: throw EllPPanic("primal phase 1 should never be unbounded");
    case SolutionStatus::MaxIter:
        res.kind = SolverResult::MaxIter;
        res.max_iter_obj = std::numeric_limits<double>::infinity();
        return res;
    }
    PrimalPhase2 phase_2 = PrimalPhase2::from_phase1(std::move(phase_1));
    SolutionStatus s2;
    if (resident) {
        Flat f2 = flatten(phase_2.std_form, phase_2.point);
        char err[512] = {0};
        const ellp_status rs = ellp_engine_rephase(eng.e, phase_2.std_form.c.data(), f2.kind.data(), f2.lb.data(),
                                                   f2.ub.data(), err, sizeof(err));
        if (rs != ELLP_OPTIMAL) to_status(rs, err);
        s2 = run_resident(eng.e, max_iter_, f2, phase_2.point, &res.iters_phase2);
    } else {
        s2 = solve_with_initial(phase_2.std_form, phase_2.point, &res.iters_phase2);
    }
    switch (s2) {
    case SolutionStatus::Optimal:
        res.kind = SolverResult::Optimal;
        res.solution = Solution{std::move(phase_2.std_form), std::move(phase_2.point)};
        return res;
    case SolutionStatus::Infeasible: throw EllPPanic("primal phase 2 should never be infeasible");
    case SolutionStatus::Unbounded: res.kind = SolverResult::Unbounded; return res;
    case SolutionStatus::MaxIter:
        res.kind = SolverResult::MaxIter;
        res.max_iter_obj = phase_2.obj();
        return res;
    }
    return res;
}

// dual_simplex_solver.rs:110-335
SolutionStatus DualSimplexSolver::solve_with_initial(const StandardForm &sf, DualFeasiblePoint &dp,
                                                     std::uint64_t *iters) const {
    if (iters) *iters = 0;
    Point &pt = dp.point;
    if (sf.rows() == 0) {  // :132-136
        if (!pt.B.empty()) throw EllPPanic("assertion failed: B.is_empty()");
        return solve_trivial_problem(sf, pt.x, pt.N, true);
    }
    Flat f = flatten(sf, pt);
    const ellp_opts o = make_opts(max_iter_, engine_);
    ellp_stats st{};
    char err[512] = {0};
    const ellp_status s = ellp_dual_solve_with_initial(
        static_cast<std::int64_t>(sf.rows()), static_cast<std::int64_t>(sf.cols()),
        static_cast<std::int64_t>(sf.bounds.size()), sf.A.a.data(), sf.c.data(), sf.b.data(), f.kind.data(),
        f.lb.data(), f.ub.data(), pt.x.data(), f.B.data(), static_cast<std::int64_t>(f.B.size()), f.N.data(),
        f.Nb.data(), static_cast<std::int64_t>(f.N.size()), dp.y.data(), dp.d.data(), &o, &st, err, sizeof(err));
    if (iters) *iters = st.iters;
    const SolutionStatus out = to_status(s, err);
    unflatten(f, pt);
    return out;
}

namespace {
// one phase of the dual method on a resident engine: run, bring the point and the duals back
SolutionStatus run_resident_dual(ellp_engine *e, std::uint64_t max_iter, Flat &f, DualFeasiblePoint &dp, std::uint64_t *iters) {
    ellp_stats st{};
    char err[512] = {0};
    const ellp_status s = ellp_engine_run(e, max_iter, &st, err, sizeof(err));
    if (iters) *iters = st.iters;
    SolutionStatus out = to_status(s, err);
    const ellp_status rs = ellp_engine_read_point(e, dp.point.x.data(), f.B.data(), f.N.data(), f.Nb.data(), dp.y.data(),
                                                  dp.d.data(), err, sizeof(err));
    if (rs != ELLP_OPTIMAL) out = to_status(rs, err);
    unflatten(f, dp.point);
    return out;
}
}  // namespace

// dual_simplex_solver.rs:33-108.  Where the engine is of the explicit-inverse kind (m > 128) and the box
// problem of phase 1 has the original standard form's matrix (no TwoSided / Fixed variable was dropped,
// dual_problem.rs:96-112), the two solve_with_initial calls are two slices of ONE resident engine:
// DualPhase2::from(phase_1) (dual_problem.rs:258-404) is done on the device from the resident B^-1
// (ellp_engine_dual_rephase) — the matrix is uploaded once and no LU is computed on the host.
SolverResult DualSimplexSolver::solve(Problem prob) const {
    SolverResult res;
    Problem orig_for_fallback = prob;  // phase_1.into_orig_prob()
    // ELLP_HOST_DUAL_POINT=1 (diagnostics): the starting point of phase 1 is made on the host, as the reference makes it
    const char *hostpt = std::getenv("ELLP_HOST_DUAL_POINT");
    auto p1 = DualPhase1::from_problem(std::move(prob), /*defer_point=*/!(hostpt && hostpt[0] == '1'));
    if (!p1) { res.kind = SolverResult::Infeasible; return res; }
    DualPhase1 phase_1 = std::move(*p1);
    const StandardForm &sf1 = phase_1.std_form;
    const bool deferred = phase_1.point_deferred;  // rows > 128, N not empty: the device makes y, d, x (:162-214)
    const bool resident = sf1.rows() > 128 && !phase_1.point.point.N.empty() &&
                          sf1.rows() == phase_1.orig_std_form.rows() && sf1.cols() == phase_1.orig_std_form.cols() &&
                          sf1.bounds.size() == phase_1.orig_std_form.bounds.size() && sf1.A.a == phase_1.orig_std_form.A.a;
    EngineHandle eng;
    SolutionStatus s1;
    if (deferred) {
        Flat f1 = flatten(sf1, phase_1.point.point);
        const ellp_opts o = make_opts(max_iter_, engine_);
        char err[512] = {0};
        const ellp_status cs = ellp_engine_create_dual_phase1(
            static_cast<std::int64_t>(sf1.rows()), static_cast<std::int64_t>(sf1.cols()), sf1.A.a.data(), sf1.c.data(),
            sf1.b.data(), f1.kind.data(), f1.lb.data(), f1.ub.data(), f1.B.data(), f1.N.data(), &o, &eng.e, err,
            sizeof(err));
        if (cs != ELLP_OPTIMAL) to_status(cs, err);
        s1 = run_resident_dual(eng.e, max_iter_, f1, phase_1.point, &res.iters_phase1);
    } else if (resident) {
        Flat f1 = flatten(sf1, phase_1.point.point);
        const ellp_opts o = make_opts(max_iter_, engine_);
        char err[512] = {0};
        const ellp_status cs = ellp_engine_create(
            ELLP_ENGINE_DUAL, static_cast<std::int64_t>(sf1.rows()), static_cast<std::int64_t>(sf1.cols()),
            static_cast<std::int64_t>(sf1.bounds.size()), sf1.A.a.data(), sf1.c.data(), sf1.b.data(), f1.kind.data(),
            f1.lb.data(), f1.ub.data(), phase_1.point.point.x.data(), f1.B.data(), static_cast<std::int64_t>(f1.B.size()),
            f1.N.data(), f1.Nb.data(), static_cast<std::int64_t>(f1.N.size()), phase_1.point.y.data(),
            phase_1.point.d.data(), &o, &eng.e, err, sizeof(err));
        if (cs != ELLP_OPTIMAL) to_status(cs, err);
        s1 = run_resident_dual(eng.e, max_iter_, f1, phase_1.point, &res.iters_phase1);
    } else {
        s1 = solve_with_initial(phase_1.std_form, phase_1.point, &res.iters_phase1);
    }
    switch (s1) {
    case SolutionStatus::Optimal: {
        const double obj = phase_1.obj();
        if (!(obj < EPS)) throw EllPPanic("assertion failed: obj < EPS");
        if (!(obj > -EPS)) {
            // dual infeasible: classify with the primal solver, default max_iter (dual…:51-66)
            SolverResult r = PrimalSimplexSolver().with_engine(engine_).solve(std::move(orig_for_fallback));
            if (r.kind == SolverResult::Optimal)
                throw EllPPanic("assertion failed: matches!(result, Infeasible | Unbounded | MaxIter)");
            return r;
        }
        break;
    }
    case SolutionStatus::Infeasible: throw EllPPanic("dual phase 1 should never be infeasible");
    case SolutionStatus::Unbounded: throw EllPPanic("dual phase 1 should never be unbounded");
    case SolutionStatus::MaxIter:
        res.kind = SolverResult::MaxIter;
        res.max_iter_obj = std::numeric_limits<double>::infinity();
        return res;
    }
    DualPhase2 phase_2;
    SolutionStatus s2;
    if (resident) {
        phase_2 = DualPhase2::shell_from_phase1(std::move(phase_1));
        Flat f2 = flatten(phase_2.std_form, phase_2.point.point);
        char err[512] = {0};
        // an engine that runs whole iterations inside one persistent launch (m <= 128, pipeline 3, or a certified-hybrid engine
        // that has repeated its phase with the exact kernel) keeps no inverse: no hand-off on the device there — asked of
        // the engine itself (ELLP_TAP_STATE [19] = launches per iteration, 0 for that kind), not inferred from an error code
        double tapv[20] = {0};
        const bool has_inverse = ellp_engine_tap(eng.e, ELLP_TAP_STATE, tapv, 20) >= 20 && tapv[19] != 0.0;
        const ellp_status rs = has_inverse ? ellp_engine_dual_rephase(eng.e, phase_2.std_form.c.data(), phase_2.std_form.b.data(),
                                                                      f2.kind.data(), f2.lb.data(), f2.ub.data(), err, sizeof(err))
                                           : ELLP_ERR_ARG;
        if (rs == ELLP_OPTIMAL) {
            s2 = run_resident_dual(eng.e, max_iter_, f2, phase_2.point, &res.iters_phase2);
        } else if (!has_inverse || rs == ELLP_ERR_PANIC) {
            // no hand-off on this engine, or one of the reference's EPS
            // assertions on the sign of d tripped on the device's inverse, which is not the fresh LU the reference
            // takes (dual_problem.rs:275-284): do the hand-off as the reference does, on the host — if the assertion
            // is the reference's own it fires again there, as the panic it is.  Any other error of the hand-off (a device
            // error, bad arguments) is reported, not retried
            ellp_engine_destroy(eng.e);
            eng.e = nullptr;
            phase_2.point_on_host();
            s2 = solve_with_initial(phase_2.std_form, phase_2.point, &res.iters_phase2);
        } else {
            to_status(rs, err);
            return res;
        }
    } else {
        if (eng.e) {  // phase 1 ran on a resident engine whose matrix is not phase 2's
            ellp_engine_destroy(eng.e);
            eng.e = nullptr;
        }
        phase_2 = DualPhase2::from_phase1(std::move(phase_1));
        s2 = solve_with_initial(phase_2.std_form, phase_2.point, &res.iters_phase2);
    }
    switch (s2) {
    case SolutionStatus::Optimal:
        res.kind = SolverResult::Optimal;
        res.solution = Solution{std::move(phase_2.std_form), std::move(phase_2.point.point)};
        return res;
    case SolutionStatus::Infeasible: res.kind = SolverResult::Infeasible; return res;
    case SolutionStatus::Unbounded: throw EllPPanic("dual phase 2 should never return unbounded");
    case SolutionStatus::MaxIter:
        res.kind = SolverResult::MaxIter;
        res.max_iter_obj = phase_2.obj();
        return res;
    }
    return res;
}

}  // namespace ellp
