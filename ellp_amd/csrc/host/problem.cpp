// problem.cpp — Problem builder + StandardForm (src/problem.rs, src/standard_form.rs).
#include <cmath>
#include <cstdlib>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "ellp.h"

namespace ellp {

namespace {

using dense::Index;

// `A.transpose().col_piv_qr()` (standard_form.rs:142).  Large matrices go to the device
// (ellp_hip_qr_transposed: the same steps with the same pivot rule, sums reduced in parallel; bitwise the
// host loop with ELLP_QR_EXACT=1 — ~2 n m^2 flop are a minute of one host core at m=2000, n=7000); small
// ones, and any matrix when no HIP device is present, use the host loop.  ELLP_QR_DEVICE=1 / 0 forces
// the choice (tests).
dense::ColPivQR col_piv_qr_of_transpose(const dense::Matrix &A) {
    const char *force = std::getenv("ELLP_QR_DEVICE");
    const bool want_device = force ? force[0] == '1' : (A.rows * A.cols >= (Index)1 << 20);
    if (want_device && !A.is_empty()) {
        const Index mn = std::min(A.rows, A.cols);
        std::vector<std::int64_t> piv(static_cast<size_t>(mn));
        dense::ColPivQR qr;
        qr.r_diag_abs.assign(static_cast<size_t>(mn), 0.0);
        char err[256] = {0};
        const ellp_status s = ellp_hip_qr_transposed(static_cast<std::int64_t>(A.rows), static_cast<std::int64_t>(A.cols),
                                                     A.a.data(), piv.data(), qr.r_diag_abs.data(), -1, err, sizeof(err));
        if (s == ELLP_OPTIMAL) {
            for (Index i = 0; i < mn; ++i) qr.p.append_permutation(i, static_cast<Index>(piv[static_cast<size_t>(i)]));
            return qr;
        }
        if (force && force[0] == '1') throw std::runtime_error(std::string("device QR failed: ") + err);
    }
    return dense::ColPivQR(A.transpose());
}

}  // namespace


using dense::Index;

// src/problem.rs:24-33
VariableId Problem::add_var(double obj_coeff, Bound bound, std::optional<std::string> name) {
    return add_var_with_id(obj_coeff, bound, variables.size(), std::move(name));
}

// src/problem.rs:35-83
VariableId Problem::add_var_with_id(double obj_coeff, Bound bound, VariableId id, std::optional<std::string> name) {
    if (bound.kind == Bound::TwoSided && bound.lb > bound.ub) {
        std::ostringstream os;
        os << "invalid variable bounds: (" << bound.lb << ", " << bound.ub << ")";
        throw EllPError(os.str());
    }
    bool valid = true;
    switch (bound.kind) {
    case Bound::Free: break;
    case Bound::Lower: valid = std::isfinite(bound.lb); break;
    case Bound::Upper: valid = std::isfinite(bound.ub); break;
    case Bound::TwoSided: valid = std::isfinite(bound.lb) && std::isfinite(bound.ub); break;
    case Bound::Fixed: valid = std::isfinite(bound.lb); break;
    }
    if (!valid) throw EllPError("invalid bound");
    if (name && !var_names_.insert(*name).second)
        throw EllPError("variable names must be unique, " + *name + " was added twice");
    variables.push_back(Variable{id, obj_coeff, bound, std::move(name)});
    if (!var_ids_.insert(id).second) {
        std::ostringstream os;
        os << "cannot add variable with VariableId(" << id << "), that id is already used";
        throw EllPError(os.str());
    }
    return id;
}

// src/problem.rs:85-106
void Problem::add_constraint(std::vector<std::pair<VariableId, double>> coeffs, ConstraintOp op, double rhs) {
    for (const auto &c : coeffs)
        if (!var_ids_.count(c.first)) {
            std::ostringstream os;
            os << "VariableId(" << c.first << ") is invalid";
            throw EllPError(os.str());
        }
    constraints.push_back(Constraint{std::move(coeffs), op, rhs});
}

// src/problem.rs:108-153, :237-250
bool Problem::is_feasible(const std::vector<double> &x) const {
    if (x.size() != variables.size()) return false;
    for (size_t i = 0; i < variables.size(); ++i) {
        const Bound &b = variables[i].bound;
        const double v = x[i];
        switch (b.kind) {
        case Bound::Free: break;
        case Bound::Lower: if (v < b.lb - EPS) return false; break;
        case Bound::Upper: if (v > b.ub + EPS) return false; break;
        case Bound::TwoSided: if (v < b.lb - EPS || v > b.ub + EPS) return false; break;
        case Bound::Fixed: if (std::fabs(v - b.lb) > EPS) return false; break;
        }
    }
    for (const auto &con : constraints) {
        double lhs = 0.0;
        for (const auto &c : con.coeffs) lhs += c.second * x[c.first];
        bool ok = true;
        switch (con.op) {
        case ConstraintOp::Lte: ok = lhs <= con.rhs + EPS; break;
        case ConstraintOp::Eq: ok = std::fabs(lhs - con.rhs) < EPS; break;
        case ConstraintOp::Gte: ok = lhs >= con.rhs - EPS; break;
        }
        if (!ok) return false;
    }
    return true;
}

// src/standard_form.rs:48
double StandardForm::obj(const std::vector<double> &x) const {
    double s = 0.0;
    for (size_t i = 0; i < c.size(); ++i) s += c[i] * x[i];
    return s;
}

// src/standard_form.rs:52-68
double StandardForm::dual_obj(const std::vector<double> &y, const std::vector<double> &d) const {
    if (d.size() != bounds.size()) throw EllPPanic("assertion failed: d.len() == self.bounds.len()");
    double obj = 0.0;
    for (size_t i = 0; i < b.size(); ++i) obj += b[i] * y[i];
    for (size_t i = 0; i < bounds.size(); ++i) {
        const Bound &bd = bounds[i];
        switch (bd.kind) {
        case Bound::Free: break;
        case Bound::Lower: obj += bd.lb * d[i]; break;
        case Bound::Upper: obj += bd.ub * d[i]; break;
        case Bound::TwoSided: obj += (d[i] > 0.0) ? bd.lb * d[i] : bd.ub * d[i]; break;
        case Bound::Fixed: obj += bd.lb * d[i]; break;
        }
    }
    return obj;
}

// src/standard_form.rs:71-74
std::vector<double> StandardForm::extract_solution(const Point &point) const {
    return std::vector<double>(point.x.begin(), point.x.begin() + static_cast<std::ptrdiff_t>(prob.variables.size()));
}

// src/standard_form.rs:78-191
std::optional<StandardForm> StandardForm::from_problem(Problem prob) {
    const Index n = static_cast<Index>(prob.variables.size());
    const Index m = static_cast<Index>(prob.constraints.size());
    Index num_slack = 0;
    for (const auto &con : prob.constraints)
        if (con.op != ConstraintOp::Eq) ++num_slack;
    const Index total_vars = n + num_slack;

    std::vector<double> c(static_cast<size_t>(total_vars), 0.0);
    dense::Matrix A(m, total_vars, 0.0);
    std::vector<double> b(static_cast<size_t>(m), 0.0);
    std::vector<Bound> bounds(static_cast<size_t>(total_vars), Bound::lower(0.0));  // slack bounds
    std::unordered_map<VariableId, Index> id_to_index;
    id_to_index.reserve(prob.variables.size());
    for (Index i = 0; i < n; ++i) {
        const Variable &v = prob.variables[static_cast<size_t>(i)];
        c[static_cast<size_t>(i)] = v.obj_coeff;
        bounds[static_cast<size_t>(i)] = v.bound;
        id_to_index[v.id] = i;
    }
    Index cur_slack_col = total_vars > 0 ? total_vars - 1 : 0;
    for (Index i = 0; i < m; ++i) {
        const Constraint &con = prob.constraints[static_cast<size_t>(i)];
        b[static_cast<size_t>(i)] = con.rhs;
        if (con.coeffs.empty() && con.rhs != 0.0) return std::nullopt;
        for (const auto &cf : con.coeffs) A(i, id_to_index.at(cf.first)) = cf.second;
        if (con.op != ConstraintOp::Eq) {
            A(i, cur_slack_col) = (con.op == ConstraintOp::Lte) ? 1.0 : -1.0;
            cur_slack_col -= 1;
        }
    }

    // remove redundant rows (:142-181)
    dense::ColPivQR qr = col_piv_qr_of_transpose(A);
    const Index r_rows = std::min(total_vars, m);
    std::vector<double> &rd = qr.r_diag_abs;
    qr.p.inv_permute_rows(b);
    for (auto &v : rd)
        if (std::fabs(v) < EPS) v = 0.0;
    const bool r_nonempty = r_rows > 0 && m > 0;
    const bool is_trivial = r_nonempty && std::fabs(rd[0]) < EPS && std::fabs(b[0]) < EPS;
    bool solvable = true;
    if (!is_trivial)
        for (Index i = 0; i < r_rows; ++i)
            if (rd[static_cast<size_t>(i)] == 0.0) { solvable = false; break; }  // tr_solve_upper_triangular -> None
    qr.p.permute_rows(b);
    if (!solvable) return std::nullopt;

    Index num_indep_rows = r_rows;
    for (Index i = 0; i < r_rows; ++i)
        if (std::fabs(rd[static_cast<size_t>(i)]) < EPS) { num_indep_rows = i; break; }
    std::vector<Index> indep_rows(static_cast<size_t>(m));
    for (Index i = 0; i < m; ++i) indep_rows[static_cast<size_t>(i)] = i;
    qr.p.permute_rows(indep_rows);
    indep_rows.resize(static_cast<size_t>(num_indep_rows));

    StandardForm sf;
    sf.A = dense::Matrix(num_indep_rows, total_vars, 0.0);
    sf.b.resize(static_cast<size_t>(num_indep_rows));
    for (Index k = 0; k < num_indep_rows; ++k) {
        const Index src = indep_rows[static_cast<size_t>(k)];
        for (Index j = 0; j < total_vars; ++j) sf.A(k, j) = A(src, j);
        sf.b[static_cast<size_t>(k)] = b[static_cast<size_t>(src)];
    }
    sf.c = std::move(c);
    sf.bounds = std::move(bounds);
    sf.prob = std::move(prob);
    return sf;
}

// solvers/trivial/solve_trivial_problem.rs:5-96 (quirk Q8 kept as is)
SolutionStatus solve_trivial_problem(const StandardForm &sf, std::vector<double> &x, std::vector<Nonbasic> &N,
                                     bool minimize) {
    N.clear();
    if (sf.c.size() != sf.bounds.size()) throw EllPPanic("assertion failed: c.len() == bounds.len()");
    const size_t len = std::min(x.size(), std::min(sf.c.size(), sf.bounds.size()));
    for (size_t i = 0; i < len; ++i) {
        const double c_i = sf.c[i];
        const Bound &bd = sf.bounds[i];
        switch (bd.kind) {
        case Bound::Free:
            N.push_back({i, NonbasicBound::Free});
            if (c_i != 0.0) return SolutionStatus::Unbounded;
            x[i] = 0.0;
            break;
        case Bound::Lower:
            N.push_back({i, NonbasicBound::Lower});
            if (c_i > 0.0) {
                if (!minimize) return SolutionStatus::Unbounded;
                x[i] = bd.lb;
            } else {
                if (!minimize && c_i != 0.0) return SolutionStatus::Unbounded;
                x[i] = bd.lb;
            }
            break;
        case Bound::Upper:
            N.push_back({i, NonbasicBound::Upper});
            if (c_i > 0.0) {
                if (minimize) return SolutionStatus::Unbounded;
                x[i] = bd.ub;
            } else {
                if (minimize && c_i != 0.0) return SolutionStatus::Unbounded;
                x[i] = bd.ub;
            }
            break;
        case Bound::TwoSided:
            if ((c_i > 0.0) == minimize) {
                N.push_back({i, NonbasicBound::Lower});
                x[i] = bd.lb;
            } else {
                N.push_back({i, NonbasicBound::Upper});
                x[i] = bd.ub;
            }
            break;
        case Bound::Fixed:
            N.push_back({i, NonbasicBound::Lower});
            x[i] = bd.lb;
            break;
        }
    }
    return SolutionStatus::Optimal;
}

}  // namespace ellp
