// phases.cpp — phase construction (once-per-solve host setup):
//   src/solvers/primal/primal_problem.rs:80-291, src/solvers/dual/dual_problem.rs:89-404
#include <cmath>
#include <cstdlib>

#include "ellp.h"
#include "ellp_hip.h"

namespace ellp {

using dense::Index;

// primal_problem.rs:80-261
std::optional<PrimalPhase1> PrimalPhase1::from_problem(Problem prob) {
    auto sf_opt = StandardForm::from_problem(std::move(prob));
    if (!sf_opt) return std::nullopt;
    StandardForm std_form = std::move(*sf_opt);
    const Index n = static_cast<Index>(std_form.cols());
    const Index m = static_cast<Index>(std_form.rows());
    std::vector<Nonbasic> N;
    std::vector<Basic> B;
    std::vector<double> v(static_cast<size_t>(n), 0.0);

    for (Index i = 0; i < n; ++i) {
        const Bound &bd = std_form.bounds[static_cast<size_t>(i)];
        switch (bd.kind) {
        case Bound::Free: break;  // set later
        case Bound::Lower: v[i] = bd.lb; N.push_back({static_cast<size_t>(i), NonbasicBound::Lower}); break;
        case Bound::Upper: v[i] = bd.ub; N.push_back({static_cast<size_t>(i), NonbasicBound::Upper}); break;
        case Bound::TwoSided: v[i] = bd.lb; N.push_back({static_cast<size_t>(i), NonbasicBound::Lower}); break;
        case Bound::Fixed: v[i] = bd.lb; N.push_back({static_cast<size_t>(i), NonbasicBound::Lower}); break;
        }
    }
    for (auto &ci : std_form.c) ci = 0.0;
    std_form.c.resize(static_cast<size_t>(n + m), 1.0);

    std::vector<Index> free_vars;
    for (Index i = 0; i < n; ++i)
        if (std_form.bounds[static_cast<size_t>(i)].kind == Bound::Free) free_vars.push_back(i);

    if (!free_vars.empty() && !std_form.A.is_empty()) {
        dense::FullPivLU lu(std_form.A.select_columns(free_vars));
        const Index nf = static_cast<Index>(free_vars.size());
        const Index max_rank = std::min(m, nf);
        Index rank = nf;  // unwrap_or_else(|| free_vars.len())
        for (Index i = 0; i < max_rank; ++i)
            if (std::fabs(lu.at(i, i)) < EPS) { rank = i; break; }
        if (rank > max_rank) throw EllPPanic("matrix slicing out of bounds (free-variable rank > rows)");
        lu.q().permute_rows(free_vars);
        for (Index k = 0; k < rank; ++k) B.push_back({static_cast<size_t>(free_vars[k])});
        for (Index k = rank; k < nf; ++k) {
            const Index i = free_vars[static_cast<size_t>(k)];
            std_form.bounds[static_cast<size_t>(i)] = Bound::fixed(0.0);
            N.push_back({static_cast<size_t>(i), NonbasicBound::Lower});
        }
        std::vector<double> b_tilde = dense::b_minus_Av(std_form.A, v, std_form.b);
        lu.p().permute_rows(b_tilde);
        // L (unit lower) then U on the leading rank x rank blocks, column-oriented
        for (Index i = 0; i < rank; ++i) {
            const double coeff = b_tilde[static_cast<size_t>(i)];
            for (Index r = i + 1; r < rank; ++r) b_tilde[r] = (-coeff) * lu.at(r, i) + b_tilde[r];
        }
        for (Index i = rank - 1; i >= 0; --i) {
            const double diag = lu.at(i, i);
            if (diag == 0.0) throw EllPPanic("called `Option::unwrap()` on a `None` value");
            const double coeff = b_tilde[static_cast<size_t>(i)] / diag;
            b_tilde[static_cast<size_t>(i)] = coeff;
            for (Index r = 0; r < i; ++r) b_tilde[r] = (-coeff) * lu.at(r, i) + b_tilde[r];
        }
        for (Index k = 0; k < rank; ++k) v[static_cast<size_t>(free_vars[k])] = b_tilde[static_cast<size_t>(k)];

        std::vector<Index> rows(static_cast<size_t>(m));
        for (Index i = 0; i < m; ++i) rows[static_cast<size_t>(i)] = i;
        lu.p().permute_rows(rows);

        b_tilde = dense::b_minus_Av(std_form.A, v, std_form.b);
        v.resize(static_cast<size_t>(n + m), 0.0);
        const Index nart = m - rank;
        std_form.A.resize_horizontally(n + nart, 0.0);
        Index cur_col = std_form.A.cols - 1;
        for (Index k = rank; k < m; ++k) {
            const Index i = rows[static_cast<size_t>(k)];
            v[static_cast<size_t>(cur_col)] = std::fabs(b_tilde[static_cast<size_t>(i)]);
            std_form.A(i, cur_col) = dense::rust_signum(b_tilde[static_cast<size_t>(i)]);
            B.push_back({static_cast<size_t>(cur_col)});
            cur_col -= 1;
        }
    } else {
        const std::vector<double> b_tilde = dense::b_minus_Av(std_form.A, v, std_form.b);
        v.resize(static_cast<size_t>(n + m), 0.0);
        std_form.A.resize_horizontally(n + m, 0.0);
        for (Index i = 0; i < m; ++i) {
            const Index index = n + i;
            v[static_cast<size_t>(index)] = std::fabs(b_tilde[static_cast<size_t>(i)]);
            std_form.A(i, index) = dense::rust_signum(b_tilde[static_cast<size_t>(i)]);
            B.push_back({static_cast<size_t>(index)});
        }
    }
    std::vector<size_t> phase_1_vars;
    for (Index i = 0; i < m; ++i) {
        phase_1_vars.push_back(std_form.bounds.size());
        std_form.bounds.push_back(Bound::lower(0.0));
    }
    PrimalPhase1 p1;
    p1.std_form = std::move(std_form);
    p1.point = Point{std::move(v), std::move(N), std::move(B)};
    p1.phase_1_vars = std::move(phase_1_vars);
    return p1;
}

// primal_problem.rs:263-291
PrimalPhase2 PrimalPhase2::from_phase1(PrimalPhase1 phase_1) {
    StandardForm std_form = std::move(phase_1.std_form);
    for (size_t i : phase_1.phase_1_vars) {
        std_form.c[i] = 0.0;
        std_form.bounds[i] = Bound::fixed(0.0);
    }
    for (size_t i = 0; i < std_form.prob.variables.size(); ++i) {
        std_form.c[i] = std_form.prob.variables[i].obj_coeff;
        std_form.bounds[i] = std_form.prob.variables[i].bound;  // undo Fixed(0) on free variables
    }
    Point point = std::move(phase_1.point);
    for (auto &var : point.N)
        if (std_form.bounds[var.index].kind == Bound::Free) var.bound = NonbasicBound::Free;
    return PrimalPhase2{std::move(std_form), std::move(point)};
}

namespace {
// y = A_B^-T c_B ; d = c - A^T y     (dual_problem.rs:162-172, :276-283)
void duals_for_basis(const StandardForm &sf, const std::vector<Basic> &B, const dense::LU &lu,
                     std::vector<double> &y, std::vector<double> &d) {
    y.resize(B.size());
    for (size_t i = 0; i < B.size(); ++i) y[i] = sf.c[B[i].index];
    if (!lu.solve_transposed(y)) throw EllPPanic("called `Option::unwrap()` on a `None` value");
    d.assign(static_cast<size_t>(sf.A.cols), 0.0);
    for (Index j = 0; j < sf.A.cols; ++j) {
        const double *cj = sf.A.col(j);
        double dot = 0.0;
        for (Index i = 0; i < sf.A.rows; ++i) dot += cj[i] * y[static_cast<size_t>(i)];
        d[static_cast<size_t>(j)] = sf.c[static_cast<size_t>(j)] - dot;
    }
}
std::vector<Index> basic_columns(const std::vector<Basic> &B) {
    std::vector<Index> idx;
    for (const auto &b : B) idx.push_back(static_cast<Index>(b.index));
    return idx;
}
}  // namespace

// dual_problem.rs:89-256
std::optional<DualPhase1> DualPhase1::from_problem(Problem prob, bool defer_point) {
    auto orig_opt = StandardForm::from_problem(std::move(prob));
    if (!orig_opt) return std::nullopt;
    StandardForm orig = std::move(*orig_opt);

    Problem phase_1_prob;
    std::vector<char> kept(orig.cols(), 0);
    for (size_t i = 0; i < orig.cols(); ++i) {
        std::optional<Bound> box;
        switch (orig.bounds[i].kind) {
        case Bound::Free: box = Bound::two_sided(-1.0, 1.0); break;
        case Bound::Lower: box = Bound::two_sided(0.0, 1.0); break;
        case Bound::Upper: box = Bound::two_sided(-1.0, 0.0); break;
        default: break;  // TwoSided / Fixed variables are dropped
        }
        if (box) {
            kept[i] = 1;
            phase_1_prob.add_var_with_id(orig.c[i], *box, i);
        }
    }
    for (size_t i = 0; i < orig.rows(); ++i) {
        std::vector<std::pair<VariableId, double>> coeffs;
        for (size_t j = 0; j < orig.cols(); ++j)
            if (kept[j]) coeffs.emplace_back(j, orig.A(static_cast<Index>(i), static_cast<Index>(j)));
        if (!coeffs.empty()) phase_1_prob.add_constraint(std::move(coeffs), ConstraintOp::Eq, 0.0);
    }
    auto sf_opt = StandardForm::from_problem(std::move(phase_1_prob));
    if (!sf_opt) return std::nullopt;
    StandardForm std_form = std::move(*sf_opt);

    const Index n = std_form.A.cols, m = std_form.A.rows;
    // `std_form.A.transpose().lu()` (:141), of which :143-160 use the diagonal of U and the row permutation.
    // From 2^20 entries on (or with ELLP_LU_DEVICE=1) it runs on the device, bitwise the same numbers
    // (ellp_hip_lu_transposed); the host loop is n m^2 flop on one core.
    dense::PermutationSequence lu_p;
    bool small_diag = false;
    {
        const char *force = std::getenv("ELLP_LU_DEVICE");
        const bool want_device = n >= m && m > 0 && (force ? force[0] == '1' : (m * n >= (Index)1 << 20));
        bool done = false;
        if (want_device) {
            std::vector<std::int64_t> piv(static_cast<size_t>(m));
            std::vector<double> ud(static_cast<size_t>(m));
            char err[256] = {0};
            const ellp_status s = ellp_hip_lu_transposed(static_cast<std::int64_t>(m), static_cast<std::int64_t>(n),
                                                         std_form.A.a.data(), piv.data(), ud.data(), -1, err, sizeof(err));
            if (s == ELLP_OPTIMAL) {
                for (Index i = 0; i < m; ++i) {
                    if (piv[static_cast<size_t>(i)] != i) lu_p.append_permutation(i, static_cast<Index>(piv[static_cast<size_t>(i)]));
                    if (std::fabs(ud[static_cast<size_t>(i)]) < EPS) small_diag = true;
                }
                done = true;
            } else if (force && force[0] == '1') {
                throw std::runtime_error(std::string("device LU failed: ") + err);
            }
        }
        if (!done) {
            dense::LU lu_t(std_form.A.transpose());
            small_diag = lu_t.any_small_diag(EPS);
            lu_p = lu_t.p();
        }
    }
    if (small_diag) throw EllPPanic("should always have a basis available");
    if (n < m) throw EllPPanic("index out of bounds: fewer columns than rows in the box problem");
    std::vector<Index> perm_cols(static_cast<size_t>(n));
    for (Index j = 0; j < n; ++j) perm_cols[static_cast<size_t>(j)] = j;
    lu_p.permute_rows(perm_cols);
    std::vector<Basic> B;
    std::vector<Nonbasic> N;
    for (Index i = 0; i < m; ++i) B.push_back({static_cast<size_t>(perm_cols[static_cast<size_t>(i)])});
    for (Index k = m; k < n; ++k) N.push_back({static_cast<size_t>(perm_cols[static_cast<size_t>(k)]), NonbasicBound::Lower});

    DualPhase1 p1;
    if (defer_point && m > 128 && !N.empty() && std_form.bounds.size() == static_cast<size_t>(n)) {
        for (const auto &bd : std_form.bounds)
            if (bd.kind != Bound::TwoSided && bd.kind != Bound::Fixed) throw EllPPanic("bounds should always be fixed or two-sided");
        p1.point_deferred = true;
        p1.point = DualFeasiblePoint{std::vector<double>(static_cast<size_t>(m), 0.0),
                                     std::vector<double>(std_form.bounds.size(), 0.0),
                                     Point{std::vector<double>(std_form.bounds.size(), 0.0), std::move(N), std::move(B)}};
    } else if (!B.empty()) {
        dense::LU A_B_lu(std_form.A.select_columns(basic_columns(B)));
        std::vector<double> y, d;
        duals_for_basis(std_form, B, A_B_lu, y, d);
        std::vector<double> x(std_form.bounds.size(), 0.0);
        if (d.size() != std_form.bounds.size()) throw EllPPanic("assertion failed: d.len() == bounds.len()");
        for (auto &nb : N) {
            const size_t i = nb.index;
            const Bound &bd = std_form.bounds[i];
            if (bd.kind == Bound::TwoSided) {
                if (d[i] >= 0.0) { x[i] = bd.lb; nb.bound = NonbasicBound::Lower; }
                else { x[i] = bd.ub; nb.bound = NonbasicBound::Upper; }
            } else if (bd.kind == Bound::Fixed) {
                x[i] = bd.lb;
                nb.bound = (d[i] >= 0.0) ? NonbasicBound::Lower : NonbasicBound::Upper;
            } else {
                throw EllPPanic("bounds should always be fixed or two-sided");
            }
        }
        std::vector<double> x_B = dense::b_minus_Av(std_form.A, x, std_form.b);
        if (!A_B_lu.solve(x_B)) throw EllPPanic("called `Option::unwrap()` on a `None` value");
        for (size_t i = 0; i < B.size(); ++i) x[B[i].index] = x_B[i];
        p1.point = DualFeasiblePoint{std::move(y), std::move(d), Point{std::move(x), std::move(N), std::move(B)}};
    } else {
        std::vector<double> x(N.size(), 0.0);
        if (N.size() != std_form.bounds.size()) throw EllPPanic("assertion failed: N.len() == bounds.len()");
        for (auto &nb : N) {
            nb.bound = NonbasicBound::Lower;
            const Bound &bd = std_form.bounds[nb.index];
            if (bd.kind == Bound::TwoSided || bd.kind == Bound::Fixed) x[nb.index] = bd.lb;
            else throw EllPPanic("bounds should always be fixed or two-sided");
        }
        p1.point = DualFeasiblePoint{{}, std_form.c, Point{std::move(x), std::move(N), std::move(B)}};
    }
    p1.std_form = std::move(std_form);
    p1.orig_std_form = std::move(orig);
    return p1;
}

// dual_problem.rs:258-404
DualPhase2 DualPhase2::from_phase1(DualPhase1 phase_1) {
    DualPhase2 p2 = shell_from_phase1(std::move(phase_1));
    p2.point_on_host();
    return p2;
}

void DualPhase2::point_on_host() {
    std::vector<Basic> B = std::move(point.point.B);
    std::vector<char> is_basic(std_form.cols(), 0);
    for (const auto &b : B) is_basic[b.index] = 1;
    if (!B.empty()) {
        if (static_cast<Index>(B.size()) != std_form.A.rows) throw EllPPanic("basis size does not match the row count");
        dense::LU A_B_lu(std_form.A.select_columns(basic_columns(B)));
        std::vector<double> y, d;
        duals_for_basis(std_form, B, A_B_lu, y, d);
        std::vector<Nonbasic> N;
        std::vector<double> x(static_cast<size_t>(std_form.A.cols), 0.0);
        for (size_t i = 0; i < is_basic.size(); ++i) {
            if (is_basic[i]) continue;
            const double d_i = d[i];
            const Bound &bd = std_form.bounds[i];
            switch (bd.kind) {
            case Bound::Free:
                if (!(std::fabs(d_i) < EPS)) throw EllPPanic("assertion failed: d_i.abs() < EPS");
                x[i] = 0.0; N.push_back({i, NonbasicBound::Free}); break;
            case Bound::Lower:
                if (!(d_i > -EPS)) throw EllPPanic("assertion failed: d_i > -EPS");
                x[i] = bd.lb; N.push_back({i, NonbasicBound::Lower}); break;
            case Bound::Upper:
                if (!(d_i < EPS)) throw EllPPanic("assertion failed: d_i < EPS");
                x[i] = bd.ub; N.push_back({i, NonbasicBound::Upper}); break;
            case Bound::TwoSided:
                if (d_i >= 0.0) { x[i] = bd.lb; N.push_back({i, NonbasicBound::Lower}); }
                else { x[i] = bd.ub; N.push_back({i, NonbasicBound::Upper}); }
                break;
            case Bound::Fixed: x[i] = bd.lb; N.push_back({i, NonbasicBound::Lower}); break;
            }
        }
        std::vector<double> x_B = dense::b_minus_Av(std_form.A, x, std_form.b);  // basics are still 0 in x
        if (!A_B_lu.solve(x_B)) throw EllPPanic("called `Option::unwrap()` on a `None` value");
        for (size_t i = 0; i < B.size(); ++i) x[B[i].index] = x_B[i];
        point = DualFeasiblePoint{std::move(y), std::move(d), Point{std::move(x), std::move(N), std::move(B)}};
    } else {
        std::vector<double> x_N(static_cast<size_t>(std_form.A.cols), 0.0);
        std::vector<Nonbasic> N;
        for (size_t i = 0; i < is_basic.size(); ++i) {
            if (is_basic[i]) continue;
            const Bound &bd = std_form.bounds[i];
            switch (bd.kind) {
            case Bound::Free: x_N[i] = 0.0; N.push_back({i, NonbasicBound::Free}); break;
            case Bound::Lower: x_N[i] = bd.lb; N.push_back({i, NonbasicBound::Lower}); break;
            case Bound::Upper: x_N[i] = bd.ub; N.push_back({i, NonbasicBound::Upper}); break;
            case Bound::TwoSided: x_N[i] = bd.lb; N.push_back({i, NonbasicBound::Lower}); break;
            case Bound::Fixed: x_N[i] = bd.lb; N.push_back({i, NonbasicBound::Lower}); break;
            }
        }
        point = DualFeasiblePoint{{}, std_form.c, Point{std::move(x_N), std::move(N), std::move(B)}};
    }
}

DualPhase2 DualPhase2::shell_from_phase1(DualPhase1 phase_1) {
    const Problem &phase_1_prob = phase_1.std_form.prob;
    StandardForm std_form = std::move(phase_1.orig_std_form);
    std::vector<char> is_basic(std_form.cols(), 0);
    std::vector<Basic> B;
    for (const auto &b : phase_1.point.point.B) {  // dual_problem.rs:264-274
        const size_t index = phase_1_prob.variables[b.index].id;
        is_basic[index] = 1;
        B.push_back({index});
    }
    std::vector<Nonbasic> N;
    for (size_t i = 0; i < is_basic.size(); ++i)
        if (!is_basic[i]) N.push_back({i, NonbasicBound::Lower});
    DualPhase2 p2;
    p2.point = DualFeasiblePoint{std::vector<double>(static_cast<size_t>(std_form.A.rows), 0.0),
                                 std::vector<double>(std_form.bounds.size(), 0.0),
                                 Point{std::vector<double>(std_form.bounds.size(), 0.0), std::move(N), std::move(B)}};
    p2.std_form = std::move(std_form);
    return p2;
}

}  // namespace ellp
