// dense.h — small dense f64 linear algebra for the once-per-solve host setup.
//
// The reference does its setup with nalgebra (Cargo.toml:16): `lu()`, `full_piv_lu()`,
// `col_piv_qr()`, triangular solves and PermutationSequence.  These are host-side,
// out of the hot path; this header restates the pieces the setup needs with
// nalgebra's conventions: column-major storage, partial pivoting that takes the FIRST entry
// of maximal modulus, P*A = L*U with unit L, permutations kept as transposition lists.
#pragma once

#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace ellp {
namespace dense {

using Index = std::int64_t;

// Column-major matrix, leading dimension == rows (nalgebra DMatrix).
struct Matrix {
    Index rows = 0, cols = 0;
    std::vector<double> a;
    Matrix() = default;
    Matrix(Index r, Index c, double fill = 0.0) : rows(r), cols(c), a(static_cast<size_t>(r * c), fill) {}
    double &operator()(Index i, Index j) { return a[static_cast<size_t>(i + j * rows)]; }
    double operator()(Index i, Index j) const { return a[static_cast<size_t>(i + j * rows)]; }
    double *col(Index j) { return a.data() + j * rows; }
    const double *col(Index j) const { return a.data() + j * rows; }
    bool is_empty() const { return rows == 0 || cols == 0; }
    Matrix transpose() const {
        Matrix t(cols, rows);
        for (Index j = 0; j < cols; ++j)
            for (Index i = 0; i < rows; ++i) t(j, i) = (*this)(i, j);
        return t;
    }
    Matrix select_columns(const std::vector<Index> &idx) const {
        Matrix s(rows, static_cast<Index>(idx.size()));
        for (size_t k = 0; k < idx.size(); ++k)
            for (Index i = 0; i < rows; ++i) s(i, static_cast<Index>(k)) = (*this)(i, idx[k]);
        return s;
    }
    void resize_horizontally(Index new_cols, double fill) {
        a.resize(static_cast<size_t>(rows * new_cols), fill);
        cols = new_cols;
    }
};

// nalgebra::PermutationSequence
struct PermutationSequence {
    std::vector<std::pair<Index, Index>> swaps;
    void append_permutation(Index i, Index j) { swaps.emplace_back(i, j); }
    template <typename V>
    void permute_rows(V &v) const {
        for (const auto &s : swaps) std::swap(v[static_cast<size_t>(s.first)], v[static_cast<size_t>(s.second)]);
    }
    template <typename V>
    void inv_permute_rows(V &v) const {
        for (auto it = swaps.rbegin(); it != swaps.rend(); ++it)
            std::swap(v[static_cast<size_t>(it->first)], v[static_cast<size_t>(it->second)]);
    }
};

// elimination step shared by LU and FullPivLU: scale the multipliers, rank-1 update the rest
inline void gauss_step(Matrix &m, double diag, Index i) {
    const double inv_diag = 1.0 / diag;
    double *ci = m.col(i);
    for (Index r = i + 1; r < m.rows; ++r) ci[r] *= inv_diag;
    for (Index k = i + 1; k < m.cols; ++k) {
        double *ck = m.col(k);
        const double f = -ck[i];
        if (f == 0.0) continue;
        for (Index r = i + 1; r < m.rows; ++r) ck[r] = f * ci[r] + ck[r];
    }
}

// nalgebra::linalg::LU (partial pivoting).  Works for rectangular input (dual_problem.rs:141).
class LU {
public:
    explicit LU(Matrix m) : lu_(std::move(m)) {
        const Index mn = std::min(lu_.rows, lu_.cols);
        for (Index i = 0; i < mn; ++i) {
            const double *ci = lu_.col(i);
            Index piv = i;
            double best = std::fabs(ci[i]);
            for (Index r = i + 1; r < lu_.rows; ++r) {
                const double v = std::fabs(ci[r]);
                if (v > best) { best = v; piv = r; }
            }
            const double diag = ci[piv];
            if (diag == 0.0) continue;
            if (piv != i) {
                p_.append_permutation(i, piv);
                for (Index k = 0; k < lu_.cols; ++k) std::swap(lu_(i, k), lu_(piv, k));
            }
            gauss_step(lu_, diag, i);
        }
    }
    const PermutationSequence &p() const { return p_; }
    Index min_dim() const { return std::min(lu_.rows, lu_.cols); }
    double u_diag(Index i) const { return lu_(i, i); }
    bool any_small_diag(double eps) const {
        for (Index i = 0; i < min_dim(); ++i)
            if (std::fabs(u_diag(i)) < eps) return true;
        return false;
    }
    // LU::solve: x = U^-1 L^-1 P b, false on a zero diagonal of U
    bool solve(std::vector<double> &b) const {
        const Index n = lu_.rows;
        p_.permute_rows(b);
        for (Index i = 0; i + 1 < n; ++i) {
            const double coeff = b[static_cast<size_t>(i)];
            if (coeff == 0.0) continue;
            const double *ci = lu_.col(i);
            for (Index r = i + 1; r < n; ++r) b[static_cast<size_t>(r)] = (-coeff) * ci[r] + b[static_cast<size_t>(r)];
        }
        for (Index i = n - 1; i >= 0; --i) {
            const double *ci = lu_.col(i);
            if (ci[i] == 0.0) return false;
            const double coeff = b[static_cast<size_t>(i)] / ci[i];
            b[static_cast<size_t>(i)] = coeff;
            if (coeff == 0.0) continue;
            for (Index r = 0; r < i; ++r) b[static_cast<size_t>(r)] = (-coeff) * ci[r] + b[static_cast<size_t>(r)];
        }
        return true;
    }
    // v <- A^-T v :  u().tr_solve_upper_triangular, l().tr_solve_lower_triangular, p().inv_permute_rows
    bool solve_transposed(std::vector<double> &v) const {
        const Index n = lu_.rows;
        for (Index i = 0; i < n; ++i) {
            const double *ci = lu_.col(i);
            double dot = 0.0;
            for (Index k = 0; k < i; ++k) dot += ci[k] * v[static_cast<size_t>(k)];
            v[static_cast<size_t>(i)] -= dot;
            if (ci[i] == 0.0) return false;
            v[static_cast<size_t>(i)] /= ci[i];
        }
        for (Index i = n - 1; i >= 0; --i) {
            const double *ci = lu_.col(i);
            double dot = 0.0;
            for (Index k = i + 1; k < n; ++k) dot += ci[k] * v[static_cast<size_t>(k)];
            v[static_cast<size_t>(i)] -= dot;
        }
        p_.inv_permute_rows(v);
        return true;
    }

private:
    Matrix lu_;
    PermutationSequence p_;
};

// nalgebra::linalg::FullPivLU: P*A*Q = L*U, pivot = first entry of maximal modulus of the
// trailing block (column-major scan).
class FullPivLU {
public:
    explicit FullPivLU(Matrix m) : lu_(std::move(m)) {
        const Index mn = std::min(lu_.rows, lu_.cols);
        for (Index i = 0; i < mn; ++i) {
            Index pr = i, pc = i;
            double best = std::fabs(lu_(i, i));
            for (Index j = i; j < lu_.cols; ++j)
                for (Index r = i; r < lu_.rows; ++r) {
                    const double v = std::fabs(lu_(r, j));
                    if (v > best) { best = v; pr = r; pc = j; }
                }
            const double diag = lu_(pr, pc);
            if (diag == 0.0) break;
            if (pc != i)
                for (Index r = 0; r < lu_.rows; ++r) std::swap(lu_(r, i), lu_(r, pc));
            q_.append_permutation(i, pc);
            if (pr != i) {
                p_.append_permutation(i, pr);
                for (Index k = 0; k < lu_.cols; ++k) std::swap(lu_(i, k), lu_(pr, k));
            }
            gauss_step(lu_, diag, i);
        }
    }
    const PermutationSequence &p() const { return p_; }
    const PermutationSequence &q() const { return q_; }
    double at(Index i, Index j) const { return lu_(i, j); }

private:
    Matrix lu_;
    PermutationSequence p_, q_;
};

// nalgebra::linalg::ColPivQR reduced to what standard_form.rs:142-181 consumes: the column
// transposition list and |R_ii|.  The pivot column at step i is the column that holds the entry
// of maximal modulus of the trailing block.
struct ColPivQR {
    PermutationSequence p;
    std::vector<double> r_diag_abs;
    ColPivQR() = default;  // filled by the device path (problem.cpp: col_piv_qr_of_transpose)
    explicit ColPivQR(Matrix m) {
        const Index mn = std::min(m.rows, m.cols);
        r_diag_abs.assign(static_cast<size_t>(mn), 0.0);
        std::vector<double> v;
        for (Index i = 0; i < mn; ++i) {
            Index pj = i;
            double best = std::fabs(m(i, i));
            for (Index j = i; j < m.cols; ++j)
                for (Index r = i; r < m.rows; ++r) {
                    const double val = std::fabs(m(r, j));
                    if (val > best) { best = val; pj = j; }
                }
            if (pj != i)
                for (Index r = 0; r < m.rows; ++r) std::swap(m(r, i), m(r, pj));
            p.append_permutation(i, pj);
            // Householder reflector for m[i.., i]
            const Index len = m.rows - i;
            double *x = m.col(i) + i;
            double sqn = 0.0;
            for (Index r = 0; r < len; ++r) sqn += x[r] * x[r];
            const double norm = std::sqrt(sqn);
            const double signed_norm = (x[0] < 0.0) ? -norm : norm;
            const double factor = (sqn + std::fabs(x[0]) * norm) * 2.0;
            r_diag_abs[static_cast<size_t>(i)] = norm;
            x[0] += signed_norm;
            if (factor == 0.0) continue;
            const double sf = std::sqrt(factor);
            double n2 = 0.0;
            for (Index r = 0; r < len; ++r) { x[r] /= sf; n2 += x[r] * x[r]; }
            n2 = std::sqrt(n2);
            if (n2 != 0.0)
                for (Index r = 0; r < len; ++r) x[r] /= n2;
            for (Index j = i + 1; j < m.cols; ++j) {
                double *cj = m.col(j) + i;
                double dot = 0.0;
                for (Index r = 0; r < len; ++r) dot += x[r] * cj[r];
                const double f2 = -2.0 * dot;
                for (Index r = 0; r < len; ++r) cj[r] = f2 * x[r] + cj[r];
            }
        }
    }
};

// out = b - A*v (A*v accumulated column by column)
inline std::vector<double> b_minus_Av(const Matrix &A, const std::vector<double> &v, const std::vector<double> &b) {
    std::vector<double> av(static_cast<size_t>(A.rows), 0.0);
    for (Index j = 0; j < A.cols; ++j) {
        const double vj = v[static_cast<size_t>(j)];
        if (vj == 0.0) continue;
        const double *cj = A.col(j);
        for (Index i = 0; i < A.rows; ++i) av[static_cast<size_t>(i)] += cj[i] * vj;
    }
    std::vector<double> out(static_cast<size_t>(A.rows));
    for (Index i = 0; i < A.rows; ++i) out[static_cast<size_t>(i)] = b[static_cast<size_t>(i)] - av[static_cast<size_t>(i)];
    return out;
}

inline double rust_signum(double v) {
    if (std::isnan(v)) return v;
    return std::signbit(v) ? -1.0 : 1.0;
}

}  // namespace dense
}  // namespace ellp
