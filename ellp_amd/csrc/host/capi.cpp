// capi.cpp — extern "C" surface of the host mirror (include/ellp_host.h).  Exceptions never
// cross this boundary: EllPError / panics of the reference become status codes + messages.
#include <cstdio>
#include <cstring>

#include "ellp.h"
#include "ellp_host.h"

struct ellp_problem {
    ellp::Problem prob;
};

namespace {
void put(char *buf, size_t len, const std::string &msg) {
    if (buf && len) std::snprintf(buf, len, "%s", msg.c_str());
}
ellp::Bound make_bound(int kind, double lb, double ub) {
    switch (kind) {
    case ELLP_BOUND_FREE: return ellp::Bound::free();
    case ELLP_BOUND_LOWER: return ellp::Bound::lower(lb);
    case ELLP_BOUND_UPPER: return ellp::Bound::upper(ub);
    case ELLP_BOUND_TWOSIDED: return ellp::Bound::two_sided(lb, ub);
    case ELLP_BOUND_FIXED: return ellp::Bound::fixed(lb);
    default: throw ellp::EllPError("invalid bound kind");
    }
}
}  // namespace

extern "C" {

ellp_problem *ellp_problem_new(void) { return new ellp_problem(); }
ellp_problem *ellp_problem_clone(const ellp_problem *p) { return p ? new ellp_problem(*p) : nullptr; }
void ellp_problem_free(ellp_problem *p) { delete p; }

int64_t ellp_problem_add_var_with_id(ellp_problem *p, double obj_coeff, int bound_kind, double lb, double ub,
                                     int64_t id, const char *name, char *errbuf, size_t errlen) {
    try {
        std::optional<std::string> nm;
        if (name) nm = std::string(name);
        if (id < 0) throw ellp::EllPError("negative variable id");
        return static_cast<int64_t>(p->prob.add_var_with_id(obj_coeff, make_bound(bound_kind, lb, ub),
                                                            static_cast<ellp::VariableId>(id), std::move(nm)));
    } catch (const std::exception &e) {
        put(errbuf, errlen, e.what());
        return -1;
    }
}
int64_t ellp_problem_add_var(ellp_problem *p, double obj_coeff, int bound_kind, double lb, double ub,
                             const char *name, char *errbuf, size_t errlen) {
    return ellp_problem_add_var_with_id(p, obj_coeff, bound_kind, lb, ub,
                                        static_cast<int64_t>(p->prob.variables.size()), name, errbuf, errlen);
}
int ellp_problem_add_constraint(ellp_problem *p, int64_t n, const int64_t *ids, const double *coeffs, int op,
                                double rhs, char *errbuf, size_t errlen) {
    try {
        std::vector<std::pair<ellp::VariableId, double>> cf;
        cf.reserve(static_cast<size_t>(n));
        for (int64_t k = 0; k < n; ++k) {
            if (ids[k] < 0) throw ellp::EllPError("negative variable id");
            cf.emplace_back(static_cast<ellp::VariableId>(ids[k]), coeffs[k]);
        }
        const ellp::ConstraintOp o = op == ELLP_OP_LTE ? ellp::ConstraintOp::Lte
                                     : op == ELLP_OP_GTE ? ellp::ConstraintOp::Gte
                                                         : ellp::ConstraintOp::Eq;
        p->prob.add_constraint(std::move(cf), o, rhs);
        return 0;
    } catch (const std::exception &e) {
        put(errbuf, errlen, e.what());
        return -1;
    }
}
int64_t ellp_problem_num_vars(const ellp_problem *p) { return static_cast<int64_t>(p->prob.variables.size()); }
int64_t ellp_problem_num_constraints(const ellp_problem *p) { return static_cast<int64_t>(p->prob.constraints.size()); }
int ellp_problem_is_feasible(const ellp_problem *p, const double *x, int64_t n) {
    return p->prob.is_feasible(std::vector<double>(x, x + n)) ? 1 : 0;
}
ellp_problem *ellp_parse_mps(const char *text, char *errbuf, size_t errlen) {
    try {
        auto *p = new ellp_problem();
        p->prob = ellp::parse_mps(text ? text : "");
        return p;
    } catch (const std::exception &e) {
        put(errbuf, errlen, e.what());
        return nullptr;
    }
}

int ellp_solve(const ellp_problem *p, int solver, uint64_t max_iter, const ellp_opts *opts, ellp_result *out) {
    std::memset(out, 0, sizeof(*out));
    ellp::EngineOptions eo;
    if (opts) {
        eo.device = opts->device;
        eo.refactor_period = opts->refactor_period;
        eo.btran_mode = opts->btran_mode;
        eo.poll_interval = opts->poll_interval;
        eo.pipeline = opts->pipeline;
        eo.partial_segments = opts->partial_segments;
        eo.flags = opts->flags;
    }
    const std::optional<std::uint64_t> mi =
        max_iter == ELLP_MAX_ITER_NONE ? std::nullopt : std::optional<std::uint64_t>(max_iter);
    try {
        ellp::SolverResult r = solver == ELLP_SOLVER_PRIMAL
                                   ? ellp::PrimalSimplexSolver(mi).with_engine(eo).solve(p->prob)
                                   : ellp::DualSimplexSolver(mi).with_engine(eo).solve(p->prob);
        out->iters_phase1 = r.iters_phase1;
        out->iters_phase2 = r.iters_phase2;
        switch (r.kind) {
        case ellp::SolverResult::Optimal: {
            out->status = ELLP_OPTIMAL;
            out->obj = r.solution->obj();
            const std::vector<double> x = r.solution->x();
            out->nx = static_cast<int64_t>(x.size());
            out->x = new double[x.size() ? x.size() : 1];
            std::memcpy(out->x, x.data(), sizeof(double) * x.size());
            break;
        }
        case ellp::SolverResult::Infeasible: out->status = ELLP_INFEASIBLE; break;
        case ellp::SolverResult::Unbounded: out->status = ELLP_UNBOUNDED; break;
        case ellp::SolverResult::MaxIter:
            out->status = ELLP_MAXITER;
            out->obj = r.max_iter_obj;
            break;
        }
    } catch (const ellp::EllPError &e) {
        out->status = std::strstr(e.what(), "not invertible") ? ELLP_ERR_SINGULAR : ELLP_ERR_BAD_DIMS;
        put(out->err, sizeof(out->err), e.what());
    } catch (const ellp::EllPPanic &e) {
        out->status = ELLP_ERR_PANIC;
        put(out->err, sizeof(out->err), e.what());
    } catch (const std::exception &e) {
        out->status = ELLP_ERR_DEVICE;
        put(out->err, sizeof(out->err), e.what());
    }
    return out->status;
}

static void fill_flat(ellp_flat_phase *o, const ellp::StandardForm &sf, const ellp::Point &pt,
                      const std::vector<double> *y, const std::vector<double> *d) {
    auto dupd = [](const std::vector<double> &v) {
        double *p = new double[v.size() ? v.size() : 1];
        std::memcpy(p, v.data(), sizeof(double) * v.size());
        return p;
    };
    o->m = static_cast<int64_t>(sf.rows());
    o->n = static_cast<int64_t>(sf.cols());
    o->n_c = static_cast<int64_t>(sf.bounds.size());
    o->n_B = static_cast<int64_t>(pt.B.size());
    o->n_N = static_cast<int64_t>(pt.N.size());
    o->A = dupd(sf.A.a);
    o->c = dupd(sf.c);
    o->b = dupd(sf.b);
    o->x = dupd(pt.x);
    std::vector<double> lb, ub;
    o->bound_kind = new uint8_t[sf.bounds.size() ? sf.bounds.size() : 1];
    for (size_t i = 0; i < sf.bounds.size(); ++i) {
        o->bound_kind[i] = static_cast<uint8_t>(sf.bounds[i].kind);
        lb.push_back(sf.bounds[i].lb);
        ub.push_back(sf.bounds[i].kind == ellp::Bound::Fixed ? sf.bounds[i].lb : sf.bounds[i].ub);
    }
    o->lb = dupd(lb);
    o->ub = dupd(ub);
    o->B_index = new int64_t[pt.B.size() ? pt.B.size() : 1];
    for (size_t i = 0; i < pt.B.size(); ++i) o->B_index[i] = static_cast<int64_t>(pt.B[i].index);
    o->N_index = new int64_t[pt.N.size() ? pt.N.size() : 1];
    o->N_bound = new uint8_t[pt.N.size() ? pt.N.size() : 1];
    for (size_t j = 0; j < pt.N.size(); ++j) {
        o->N_index[j] = static_cast<int64_t>(pt.N[j].index);
        o->N_bound[j] = static_cast<uint8_t>(pt.N[j].bound);
    }
    o->y = y ? dupd(*y) : nullptr;
    o->d = d ? dupd(*d) : nullptr;
}

int ellp_debug_phase1(const ellp_problem *p, int solver, ellp_flat_phase *out, char *errbuf, size_t errlen) {
    std::memset(out, 0, sizeof(*out));
    try {
        if (solver == ELLP_SOLVER_PRIMAL) {
            auto ph = ellp::PrimalPhase1::from_problem(p->prob);
            if (!ph) return 1;
            fill_flat(out, ph->std_form, ph->point, nullptr, nullptr);
        } else {
            auto ph = ellp::DualPhase1::from_problem(p->prob);
            if (!ph) return 1;
            fill_flat(out, ph->std_form, ph->point.point, &ph->point.y, &ph->point.d);
        }
        return 0;
    } catch (const std::exception &e) {
        put(errbuf, errlen, e.what());
        return ELLP_ERR_PANIC;
    }
}

void ellp_flat_phase_free(ellp_flat_phase *f) {
    if (!f) return;
    delete[] f->A; delete[] f->c; delete[] f->b; delete[] f->lb; delete[] f->ub; delete[] f->x;
    delete[] f->y; delete[] f->d; delete[] f->bound_kind; delete[] f->N_bound;
    delete[] f->B_index; delete[] f->N_index;
    std::memset(f, 0, sizeof(*f));
}

void ellp_result_free(ellp_result *r) {
    if (r && r->x) {
        delete[] r->x;
        r->x = nullptr;
    }
}

}  // extern "C"
