// ellp.h — C++ host mirror of kehlert/ellp's public API (src/lib.rs:109-129) for the one path
// this repository accelerates.  Names, argument meaning and error behaviour follow the
// reference so that tests read like tests/integration_tests.rs; the per-iteration simplex loops
// (solve_with_initial) are NOT here — they run on the MI355X through include/ellp_hip.h.
//
//   Problem / Bound / Constraint / ConstraintOp / Variable / VariableId   src/problem.rs
//   StandardForm / Point / Basic / Nonbasic / NonbasicBound               src/standard_form.rs
//   PrimalPhase1/2, DualPhase1/2                                          src/solvers/*/..._problem.rs
//   PrimalSimplexSolver / DualSimplexSolver / SolverResult / Solution     src/solvers, src/solver.rs
//   parse_mps                                                             src/parse_mps.rs
#pragma once

#include <cstdint>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "dense.h"
#include "ellp_hip.h"

namespace ellp {

constexpr double EPS = 1e-10;  // src/util.rs:1

// src/error.rs:3-11
struct EllPError : std::runtime_error {
    explicit EllPError(const std::string &msg) : std::runtime_error(msg) {}
};
// a panic!/assert! of the reference (never crosses the C boundary as an exception)
struct EllPPanic : std::logic_error {
    explicit EllPPanic(const std::string &msg) : std::logic_error(msg) {}
};

// src/problem.rs:190-197
struct Bound {
    enum Kind : std::uint8_t { Free = 0, Lower = 1, Upper = 2, TwoSided = 3, Fixed = 4 };
    Kind kind = Free;
    double lb = 0.0, ub = 0.0;
    static Bound free() { return {Free, 0.0, 0.0}; }
    static Bound lower(double l) { return {Lower, l, 0.0}; }
    static Bound upper(double u) { return {Upper, 0.0, u}; }
    static Bound two_sided(double l, double u) { return {TwoSided, l, u}; }
    static Bound fixed(double v) { return {Fixed, v, v}; }
};

using VariableId = std::size_t;  // src/problem.rs:277-278

enum class ConstraintOp { Lte, Eq, Gte };  // src/problem.rs:298-303

struct Variable {  // src/problem.rs:156-162
    VariableId id;
    double obj_coeff;
    Bound bound;
    std::optional<std::string> name;
};

struct Constraint {  // src/problem.rs:225-230
    std::vector<std::pair<VariableId, double>> coeffs;
    ConstraintOp op;
    double rhs;
};

class Problem {  // src/problem.rs:11-154
public:
    std::vector<Variable> variables;
    std::vector<Constraint> constraints;

    VariableId add_var(double obj_coeff, Bound bound, std::optional<std::string> name = std::nullopt);
    VariableId add_var_with_id(double obj_coeff, Bound bound, VariableId id,
                               std::optional<std::string> name = std::nullopt);
    void add_constraint(std::vector<std::pair<VariableId, double>> coeffs, ConstraintOp op, double rhs);
    bool is_feasible(const std::vector<double> &x) const;

private:
    std::unordered_set<std::string> var_names_;
    std::unordered_set<VariableId> var_ids_;
};

// src/standard_form.rs:205-221
enum class NonbasicBound : std::uint8_t { Lower = 0, Upper = 1, Free = 2 };
struct Nonbasic {
    std::size_t index;
    NonbasicBound bound;
};
struct Basic {
    std::size_t index;
};
struct Point {  // src/standard_form.rs:20-25
    std::vector<double> x;
    std::vector<Nonbasic> N;
    std::vector<Basic> B;
};

struct StandardForm {  // src/standard_form.rs:27-75
    std::vector<double> c;
    dense::Matrix A;
    std::vector<double> b;
    std::vector<Bound> bounds;
    Problem prob;

    std::size_t rows() const { return static_cast<std::size_t>(A.rows); }
    std::size_t cols() const { return static_cast<std::size_t>(A.cols); }
    double obj(const std::vector<double> &x) const;
    double dual_obj(const std::vector<double> &y, const std::vector<double> &d) const;
    std::vector<double> extract_solution(const Point &point) const;

    // impl From<Problem> for Option<StandardForm>  (standard_form.rs:78-191)
    static std::optional<StandardForm> from_problem(Problem prob);
};

struct DualFeasiblePoint {  // src/solvers/dual/dual_problem.rs:11-16
    std::vector<double> y, d;
    Point point;
};

struct PrimalPhase2;
struct PrimalPhase1 {  // src/solvers/primal/primal_problem.rs:38-43
    StandardForm std_form;
    Point point;
    std::vector<std::size_t> phase_1_vars;
    double obj() const { return std_form.obj(point.x); }
    static std::optional<PrimalPhase1> from_problem(Problem prob);  // :80-261
};
struct PrimalPhase2 {  // :59-63
    StandardForm std_form;
    Point point;
    double obj() const { return std_form.obj(point.x); }
    static PrimalPhase2 from_phase1(PrimalPhase1 phase_1);  // :263-291
};

struct DualPhase1 {  // src/solvers/dual/dual_problem.rs:41-46
    StandardForm std_form;
    DualFeasiblePoint point;
    StandardForm orig_std_form;
    // true: B and N are set (the LU of A^T picked them, :139-160) but y, d, x and the nonbasic labels are still
    // zero / Lower — ellp_engine_create_dual_phase1 makes them on the device (:162-214)
    bool point_deferred = false;
    double obj() const { return std_form.dual_obj(point.y, point.d); }
    // defer_point: leave :162-214 (LU of A_B, three solves, A x) to the device when the engine is of the
    // explicit-inverse kind (rows > 128) and there is a nonbasic variable
    static std::optional<DualPhase1> from_problem(Problem prob, bool defer_point = false);  // :89-256
};
struct DualPhase2 {  // :69-73
    StandardForm std_form;
    DualFeasiblePoint point;
    double obj() const { return std_form.dual_obj(point.y, point.d); }
    static DualPhase2 from_phase1(DualPhase1 phase_1);  // :258-404
    // the same without its linear algebra: original standard form, B mapped through the variable ids, N listed
    // in variable order with placeholder labels, x / y / d zero — for the device hand-off
    // (ellp_engine_dual_rephase), which fills them in
    static DualPhase2 shell_from_phase1(DualPhase1 phase_1);
    // the linear algebra of from_phase1 on a shell (dual_problem.rs:275-350: LU of A_B, y, d, labels, x): what the host
    // falls back to when the device hand-off is not available (an engine of the LU-per-iteration kind keeps no
    // inverse) or trips one of the reference's EPS assertions on its inexact inverse
    void point_on_host();
};

enum class SolutionStatus { Optimal, Infeasible, Unbounded, MaxIter };  // src/solver.rs:27-33

struct Solution {  // src/solver.rs:35-54
    StandardForm std_form;
    Point point;
    double obj() const { return std_form.obj(point.x); }
    std::vector<double> x() const { return std_form.extract_solution(point); }
};

struct SolverResult {  // src/solver.rs:6-12
    enum Kind { Optimal, Infeasible, Unbounded, MaxIter } kind = Infeasible;
    std::optional<Solution> solution;  // Optimal
    double max_iter_obj = 0.0;         // MaxIter { obj }
    std::uint64_t iters_phase1 = 0, iters_phase2 = 0;  // extension: iteration counts of the device loops
};

// solvers/trivial/solve_trivial_problem.rs:5-96
SolutionStatus solve_trivial_problem(const StandardForm &std_form, std::vector<double> &x,
                                     std::vector<Nonbasic> &N, bool minimize);

// Engine knobs that have no counterpart in the reference (0 = engine default).
struct EngineOptions {
    int device = -1;
    int refactor_period = 0;
    int btran_mode = 0;
    int poll_interval = 0;
    int partial_segments = 0;  // ellp_opts.partial_segments (> 1: partial pricing, an opt-in extension)
    int pipeline = 0;  // ellp_opts.pipeline: 0 = chosen by size, 1 / 2 = launches per iteration, 3 = persistent small kernel
    int flags = 0;     // ellp_opts.flags (include/ellp_hip.h): ELLP_FLAG_DENSE_PRICING, and the opt-in extensions ELLP_FLAG_DUAL_MAX_VIOLATION,
                       // ELLP_FLAG_PRIMAL_STEEPEST_EDGE, ELLP_FLAG_DUAL_BOUND_FLIPPING; ELLP_FLAG_NO_CERTIFY
};

class PrimalSimplexSolver {  // src/solvers/primal/primal_simplex_solver.rs:15-93
public:
    PrimalSimplexSolver() : max_iter_(1000) {}  // Default, :19-23
    explicit PrimalSimplexSolver(std::optional<std::uint64_t> max_iter)  // new(), :26-30
        : max_iter_(max_iter.value_or(std::numeric_limits<std::uint64_t>::max())) {}
    PrimalSimplexSolver &with_engine(const EngineOptions &o) { engine_ = o; return *this; }

    SolverResult solve(Problem prob) const;  // :32-93
    // :95-236 — the loop itself runs on the GPU (ellp_primal_solve_with_initial)
    SolutionStatus solve_with_initial(const StandardForm &std_form, Point &pt, std::uint64_t *iters = nullptr) const;

private:
    std::uint64_t max_iter_;
    EngineOptions engine_;
};

class DualSimplexSolver {  // src/solvers/dual/dual_simplex_solver.rs:16-108
public:
    DualSimplexSolver() : max_iter_(1000) {}
    explicit DualSimplexSolver(std::optional<std::uint64_t> max_iter)
        : max_iter_(max_iter.value_or(std::numeric_limits<std::uint64_t>::max())) {}
    DualSimplexSolver &with_engine(const EngineOptions &o) { engine_ = o; return *this; }

    SolverResult solve(Problem prob) const;
    SolutionStatus solve_with_initial(const StandardForm &std_form, DualFeasiblePoint &pt,
                                      std::uint64_t *iters = nullptr) const;

private:
    std::uint64_t max_iter_;
    EngineOptions engine_;
};

// src/parse_mps.rs:11-66.  Variables/rows are taken in file order (the reference iterates
// HashMaps, so its order is unspecified).
struct MpsParsingError : std::runtime_error {
    explicit MpsParsingError(const std::string &msg) : std::runtime_error("MPS parsing error. " + msg) {}
};
Problem parse_mps(const std::string &mps);

}  // namespace ellp
