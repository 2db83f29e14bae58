// parse_mps.cpp — free-format, one-entry-per-line MPS subset (src/parse_mps.rs:23-546):
// NAME / ROWS / COLUMNS / RHS / BOUNDS (UP, LO, FR) / ENDATA.  Same accepted grammar and error
// conditions; variables and rows keep FILE order (the reference iterates HashMaps, :29,:41, so
// its order is unspecified and its tests pin objectives only).
#include <map>
#include <sstream>

#include "ellp.h"

namespace ellp {

namespace {

struct Row {
    bool objective = false;
    ConstraintOp op = ConstraintOp::Eq;
    std::vector<std::pair<std::string, double>> coeffs;
    std::optional<double> rhs;
};
struct Col {
    double obj_coeff = 0.0;
    std::optional<Bound> bound;
};

std::vector<std::string> split_ws(const std::string &line) {
    std::istringstream is(line);
    std::vector<std::string> t;
    std::string w;
    while (is >> w) t.push_back(w);
    return t;
}
std::string trim(const std::string &s) {
    const auto a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    const auto b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}
bool starts_with(const std::string &s, const char *p) { return s.rfind(p, 0) == 0; }
double parse_f64(const std::string &tok, const char *what, const std::string &line) {
    try {
        size_t pos = 0;
        const double v = std::stod(tok, &pos);
        if (pos != tok.size()) throw std::invalid_argument(tok);
        return v;
    } catch (const std::exception &) {
        throw MpsParsingError(std::string("could not parse the ") + what + " " + tok + "\nline: " + line);
    }
}

}  // namespace

Problem parse_mps(const std::string &mps) {
    std::vector<std::string> lines;
    {
        std::istringstream is(mps);
        std::string l;
        while (std::getline(is, l))
            if (!trim(l).empty()) lines.push_back(l);
    }
    size_t at = 0;
    auto next = [&]() -> const std::string * { return at < lines.size() ? &lines[at++] : nullptr; };
    auto peek = [&]() -> const std::string * { return at < lines.size() ? &lines[at] : nullptr; };

    // NAME (:117-137)
    {
        const std::string *l = next();
        if (!l) throw MpsParsingError("could not find NAME line");
        const auto t = split_ws(*l);
        if (t.size() < 2 || t[0] != "NAME") throw MpsParsingError("could not find name in NAME line: " + *l);
    }
    // ROWS (:139-220)
    std::map<std::string, Row> rows;
    std::vector<std::string> row_order;
    {
        const std::string *l = next();
        if (!l) throw MpsParsingError("could not find ROWS line");
        if (trim(*l) != "ROWS") throw MpsParsingError("expected 'ROWS', found '" + trim(*l) + "'");
        while (const std::string *p = peek()) {
            if (starts_with(trim(*p), "COLUMNS")) break;
            ++at;
            const auto t = split_ws(*p);
            if (t.empty()) throw MpsParsingError("expected a row type character in this line: " + *p);
            Row r;
            if (t[0] == "L") r.op = ConstraintOp::Lte;
            else if (t[0] == "G") r.op = ConstraintOp::Gte;
            else if (t[0] == "E") r.op = ConstraintOp::Eq;
            else if (t[0] == "N") r.objective = true;
            else throw MpsParsingError("unexpected row type: " + t[0]);
            if (t.size() < 2) throw MpsParsingError("expected a row name in this line: " + *p);
            if (t.size() > 2) throw MpsParsingError("unexpected input in row line: " + t[2]);
            if (!rows.emplace(t[1], r).second) throw MpsParsingError("row name repeated: " + t[1]);
            row_order.push_back(t[1]);
        }
    }
    // COLUMNS (:222-331)
    std::map<std::string, Col> cols;
    std::vector<std::string> col_order;
    {
        const std::string *l = next();
        if (!l) throw MpsParsingError("could not find COLUMNS line");
        if (trim(*l) != "COLUMNS") throw MpsParsingError("expected 'COLUMNS', found '" + trim(*l) + "'");
        while (const std::string *p = peek()) {
            if (starts_with(trim(*p), "RHS")) break;
            ++at;
            const auto t = split_ws(*p);
            if (t.size() < 1) throw MpsParsingError("expected a column name in this line: " + *p);
            if (t.size() < 2) throw MpsParsingError("expected a row name in this line: " + *p);
            if (t.size() < 3) throw MpsParsingError("expected a coefficient in this line: " + *p);
            const double coeff = parse_f64(t[2], "coefficient", *p);
            if (t.size() > 3) throw MpsParsingError("unexpected input '" + t[3] + "' in column line: " + *p);
            if (!cols.count(t[0])) {
                cols[t[0]] = Col{};
                col_order.push_back(t[0]);
            }
            auto it = rows.find(t[1]);
            if (it == rows.end()) throw MpsParsingError("could not find the row " + t[1]);
            if (it->second.objective) {
                cols[t[0]].obj_coeff = coeff;
            } else {
                for (const auto &c : it->second.coeffs)
                    if (c.first == t[0])
                        throw MpsParsingError("specified constraint coefficient for the column " + t[0] +
                                              " and row " + t[1] + " more than once");
                it->second.coeffs.emplace_back(t[0], coeff);
            }
        }
    }
    // RHS (:333-425)
    {
        const std::string *l = next();
        if (!l) throw MpsParsingError("could not find RHS line");
        if (trim(*l) != "RHS") throw MpsParsingError("expected 'RHS', found '" + trim(*l) + "'");
        while (const std::string *p = peek()) {
            const std::string tl = trim(*p);
            if (starts_with(tl, "BOUNDS") || starts_with(tl, "ENDATA")) break;
            ++at;
            auto t = split_ws(*p);
            if (t.size() == 3) t.erase(t.begin());  // skip the RHS-set name
            if (t.size() < 1) throw MpsParsingError("expected a row name in this line: " + *p);
            if (t.size() < 2) throw MpsParsingError("expected a rhs value in this line: " + *p);
            const double v = parse_f64(t[1], "rhs value", *p);
            auto it = rows.find(t[0]);
            if (it == rows.end()) throw MpsParsingError("could not find the row " + t[0]);
            if (it->second.objective) throw MpsParsingError("should not specify rhs value for the objective");
            if (it->second.rhs) throw MpsParsingError("specified rhs for " + t[0] + " more than once");
            it->second.rhs = v;
            if (t.size() > 2) throw MpsParsingError("unexpected input in column line: " + t[2]);
        }
    }
    // BOUNDS (:427-546)
    {
        const std::string *p = peek();
        bool has_bounds = true;
        if (p && *p == "ENDATA") has_bounds = false;
        if (has_bounds) {
            const std::string *l = next();
            if (!l) throw MpsParsingError("could not find BOUNDS line");
            const std::string tl = trim(*l);
            if (tl == "ENDATA") {
                --at;  // let the ENDATA check below see it
            } else if (tl != "BOUNDS") {
                throw MpsParsingError("expected 'BOUNDS', found '" + tl + "'");
            } else {
                while (const std::string *q = peek()) {
                    if (starts_with(trim(*q), "ENDATA")) break;
                    ++at;
                    const auto t = split_ws(*q);
                    if (t.size() < 1) throw MpsParsingError("expected a bound type in this line: " + *q);
                    if (t.size() < 3) throw MpsParsingError("expected a column name in this line: " + *q);
                    std::optional<double> val;
                    if (t.size() > 3) val = parse_f64(t[3], "bound value", *q);
                    Bound nb;
                    if (t[0] == "UP" && val) nb = Bound::upper(*val);
                    else if (t[0] == "LO" && val) nb = Bound::lower(*val);
                    else if (t[0] == "FR" && !val) nb = Bound::free();
                    else throw MpsParsingError("invalid bound specification: " + *q);
                    auto it = cols.find(t[2]);
                    if (it == cols.end())
                        throw MpsParsingError("found bound for the column " + t[2] + ", but it does not exist");
                    std::optional<Bound> &cur = it->second.bound;
                    if (!cur) cur = nb;
                    else if (cur->kind == Bound::Upper && nb.kind == Bound::Lower) cur = Bound::two_sided(nb.lb, cur->ub);
                    else if (cur->kind == Bound::Lower && nb.kind == Bound::Upper) cur = Bound::two_sided(cur->lb, nb.ub);
                    else throw MpsParsingError("invalid bounds for " + t[2]);
                    if (t.size() > 4) throw MpsParsingError("unexpected input in column line: " + t[4]);
                }
            }
        }
    }
    {
        const std::string *l = next();
        if (!l) throw MpsParsingError("could not find ENDATA line");
        if (trim(*l) != "ENDATA") throw MpsParsingError("expected 'ENDATA', found '" + trim(*l) + "'");
        if (const std::string *extra = next()) throw MpsParsingError("unexpected line: " + *extra);
    }

    // build the Problem (:23-66)
    Problem prob;
    std::map<std::string, VariableId> var_ids;
    for (const auto &name : col_order) {
        const Col &c = cols[name];
        var_ids[name] = prob.add_var(c.obj_coeff, c.bound.value_or(Bound::lower(0.0)), name);
    }
    for (const auto &rname : row_order) {
        const Row &r = rows[rname];
        if (r.objective) continue;
        std::vector<std::pair<VariableId, double>> coeffs;
        for (const auto &c : r.coeffs) {
            auto it = var_ids.find(c.first);
            if (it == var_ids.end())
                throw MpsParsingError("for row " + rname + ", column " + c.first + " does not exist");
            coeffs.emplace_back(it->second, c.second);
        }
        prob.add_constraint(std::move(coeffs), r.op, r.rhs.value_or(0.0));
    }
    return prob;
}

}  // namespace ellp
