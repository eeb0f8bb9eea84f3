// Host-side model: UAI reading, evidence, MAR solutions, error metrics.
// Behavioural mirror of grample's model package (citations in host_model.hpp).
#include "host_model.hpp"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>

#include "../../include/grample_hip.h"

namespace gr {

static std::string letter26(int n) { // variable.go:168-189: 0=A .. 25=Z, 26=AA
    if (n == 0) return "A";
    n++;
    std::string d;
    while (n > 0) {
        int q = (n - 1) / 26, r = (n - 1) % 26;
        d.push_back((char)('A' + r));
        n = q;
    }
    std::reverse(d.begin(), d.end());
    return d;
}

Variable Variable::New(int index, int card) {
    if (index < 0) throw Error(GR_E_INVALID, "Invalid index " + std::to_string(index) + " with card " + std::to_string(card));
    if (card < 1) throw Error(GR_E_INVALID, "Invalid card " + std::to_string(card) + " for variable " + std::to_string(index));
    Variable v;
    v.ID = index;
    v.Card = card;
    v.Name = letter26(index);
    v.Marginal.assign(card, 0.0);
    v.NormMarginal(); // all-zero marginal => uniform
    return v;
}

void Variable::NormMarginal() {
    if ((int)Marginal.size() != Card) throw Error(GR_E_INVALID, "Var " + Name + " - can not norm: Card != Len(m)");
    if (Card < 1) return;
    if (Card == 1) Marginal[0] = 1.0;
    double sum = 0;
    for (double p : Marginal) sum += p;
    const double EPS = 1e-8;
    if (std::fabs(sum - 1.0) < EPS) return;
    if (std::fabs(sum) < EPS) {
        for (double &p : Marginal) p = 1.0 / (double)Card;
        return;
    }
    for (double &p : Marginal) p /= sum;
}

void Variable::Check() const {
    if ((int)Marginal.size() != Card) throw Error(GR_E_INVALID, "Variable " + Name + " Card != len(M)");
    if (FixedVal != -1 && (FixedVal < 0 || FixedVal >= Card))
        throw Error(GR_E_INVALID, "Variable " + Name + " has fixed val " + std::to_string(FixedVal) + " but must be -1 or match card " + std::to_string(Card));
    if (Card > 0) {
        double sum = 0;
        for (double p : Marginal) sum += p;
        if (std::fabs(sum - 1.0) >= 1e-8) throw Error(GR_E_INVALID, "Variable " + Name + " has marginal dist with sum=" + std::to_string(sum));
    }
}

int Model::TotalCard() const {
    int t = 0;
    for (const auto &v : Vars) t += v.Card;
    return t;
}

int Model::CalcTabSize(const std::vector<int> &vars) const {
    long long ts = 0;
    for (size_t i = 0; i < vars.size(); ++i) {
        if (i == 0) ts = 1;
        ts *= Vars[vars[i]].Card;
        if (ts > (long long)kMaxTabSize * 64) return kMaxTabSize + 1; // saturate, caller rejects
    }
    return (int)ts;
}

Function Model::NewFunction(int index, const std::vector<int> &vars) const {
    if (index < 0) throw Error(GR_E_INVALID, "Invalid index " + std::to_string(index) + " for function");
    std::string name = "func-" + std::to_string(index);
    if (vars.empty()) throw Error(GR_E_INVALID, "Empty variable list for function " + name + " is invalid");
    int ts = CalcTabSize(vars);
    if (ts < 1) throw Error(GR_E_INVALID, "Function " + name + " is invalid - could not calculate table size");
    if (ts > kMaxTabSize)
        throw Error(GR_E_INVALID, "Function over " + std::to_string(vars.size()) + " vars has size > " + std::to_string(kMaxTabSize));
    Function f;
    f.Name = name;
    f.Vars = vars;
    f.Table.assign(ts, 0.0);
    return f;
}

int Model::CalcIndex(const Function &f, const int *values, int n) const {
    if (n != (int)f.Vars.size()) throw Error(GR_E_INVALID, "Value vector does not match variables");
    long long digit = 1, loc = 0;
    for (int i = n - 1; i >= 0; --i) {
        int val = values[i], card = Vars[f.Vars[i]].Card;
        if (val < 0 || val >= card)
            throw Error(GR_E_INVALID, "Value " + std::to_string(val) + " invalid for cardinality " + std::to_string(card) + " for var " + Vars[f.Vars[i]].Name);
        loc += digit * val;
        digit *= card;
    }
    return (int)loc;
}

double Model::Eval(const Function &f, const int *values, int n) const {
    int i = CalcIndex(f, values, n);
    if (i < 0 || i >= (int)f.Table.size()) throw Error(GR_E_INVALID, "Could not find table entry for values");
    return f.Table[i];
}

void Model::UseLogSpace(Function &f) {
    if (f.IsLog) throw Error(GR_E_INVALID, "IsLog already set - double-call detected");
    const double eps = 1e-6; // log(0) is invalid: entries below eps get eps added
    for (double &t : f.Table) {
        double v = t;
        if (v < eps) v += eps;
        t = std::log(v);
    }
    f.IsLog = true;
}

void Model::Check() const {
    if (Type != "BAYES" && Type != "MARKOV") throw Error(GR_E_INVALID, "Unknown model type " + Type);
    std::set<int> ids;
    int fix = 0;
    for (const auto &v : Vars) {
        v.Check();
        if (!ids.insert(v.ID).second) throw Error(GR_E_INVALID, "Duplicate Id " + std::to_string(v.ID) + " for Var " + v.Name);
        if (v.FixedVal > -1) fix++;
    }
    if (fix >= (int)Vars.size()) throw Error(GR_E_INVALID, "Fixed variable count is " + std::to_string(fix) + " - all vars are fixed!");
    std::set<std::string> names;
    for (const auto &f : Funcs) {
        int exp = CalcTabSize(f.Vars);
        if (exp < 1) throw Error(GR_E_INVALID, "Function " + f.Name + " is invalid - can not calculate table size");
        if (exp != (int)f.Table.size()) throw Error(GR_E_INVALID, "Function " + f.Name + " expected table size " + std::to_string(exp) + ", found " + std::to_string(f.Table.size()));
        names.insert(f.Name);
    }
    if (names.size() != Funcs.size()) throw Error(GR_E_INVALID, "There are " + std::to_string(Funcs.size()) + " funcs, but " + std::to_string(names.size()) + " names");
}

// ------------------------------------------------------------------ reading
static std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) a++;
    while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

std::string UAIReader::Preprocess(const std::string &data, const std::string &reqPrefix, int *lineCount) {
    std::string out;
    int n = 0;
    bool found = reqPrefix.empty();
    size_t pos = 0;
    while (pos <= data.size()) {
        size_t nl = data.find('\n', pos);
        if (nl == std::string::npos) nl = data.size();
        std::string ln = trim(data.substr(pos, nl - pos));
        pos = nl + 1;
        if (ln.empty() || ln[0] == 'c') continue; // blank or comment
        if (!found) {
            if (ln.compare(0, reqPrefix.size(), reqPrefix) == 0) found = true;
            else continue;
        }
        if (n) out.push_back('\n');
        out += ln;
        n++;
    }
    if (lineCount) *lineCount = n;
    return out;
}

namespace {
struct FieldReader { // model/reader.go
    std::vector<std::string> fields;
    size_t pos = 0;
    explicit FieldReader(const std::string &text) {
        std::istringstream is(text);
        std::string t;
        while (is >> t) fields.push_back(t);
    }
    const std::string &Read(const char *what) {
        if (pos >= fields.size()) throw Error(GR_E_INVALID, std::string(what) + ": EOF");
        return fields[pos++];
    }
    int ReadInt(const char *what) {
        const std::string &s = Read(what);
        errno = 0;
        char *end = nullptr;
        long long v = std::strtoll(s.c_str(), &end, 10);
        if (errno || end == s.c_str() || *end || v > INT32_MAX || v < INT32_MIN)
            throw Error(GR_E_INVALID, std::string(what) + ": invalid integer \"" + s + "\"");
        return (int)v;
    }
    // a count of items that each take at least one more field: larger than what is left is malformed input, not an
    // allocation request
    int ReadCount(const char *what) {
        const int v = ReadInt(what);
        if (v > 0 && (size_t)v > fields.size() - pos) throw Error(GR_E_INVALID, std::string(what) + ": count " + std::to_string(v) + " exceeds the remaining fields");
        return v;
    }
    double ReadFloat(const char *what) {
        const std::string &s = Read(what);
        errno = 0;
        char *end = nullptr;
        double v = std::strtod(s.c_str(), &end);
        if (end == s.c_str() || *end) throw Error(GR_E_INVALID, std::string(what) + ": invalid number \"" + s + "\"");
        return v;
    }
};
} // namespace

Model UAIReader::ReadModel(const std::string &data) {
    if (data.size() < 15) throw Error(GR_E_INVALID, "Invalid data buffer: len=" + std::to_string(data.size()) + " (<15)");
    int lines = 0;
    std::string text = Preprocess(data, "", &lines);
    if (lines < 1) throw Error(GR_E_INVALID, "No lines found in file");
    FieldReader fr(text);
    if (fr.fields.size() < 6) throw Error(GR_E_INVALID, "Invalid data: only " + std::to_string(fr.fields.size()) + " fields found (<6)");
    Model m;
    m.Type = fr.Read("Error reading UAI file on Type");
    if (m.Type != "BAYES" && m.Type != "MARKOV") throw Error(GR_E_INVALID, "Unknown model type " + m.Type);
    int varCount = fr.ReadCount("Error reading UAI file on Variable count");
    if (varCount < 1) throw Error(GR_E_INVALID, "Invalid variable count: " + std::to_string(varCount));
    m.Vars.reserve(varCount);
    for (int i = 0; i < varCount; ++i) {
        int card = fr.ReadInt("Error reading Card for var");
        if (card < 1) throw Error(GR_E_INVALID, "Invalid card " + std::to_string(card) + " for var " + std::to_string(i));
        // no function can hold such a variable (function.go:59,77-79) and Check() requires one: reject before allocating its marginal
        if (card > (1 << 23)) throw Error(GR_E_INVALID, "Invalid card " + std::to_string(card) + " for var " + std::to_string(i) + ": a function over it would have size > 8388608");
        m.Vars.push_back(Variable::New(i, card));
    }
    int funcCount = fr.ReadCount("Error reading UAI file on Clique count");
    if (funcCount < 1) throw Error(GR_E_INVALID, "Invalid Clique count count: " + std::to_string(funcCount));
    m.Funcs.reserve(funcCount);
    for (int i = 0; i < funcCount; ++i) {
        int k = fr.ReadCount("Error reading Clique size");
        if (k < 1) throw Error(GR_E_INVALID, "Invalid variable count (<1) for Clique " + std::to_string(i));
        std::vector<int> fv(k);
        for (int j = 0; j < k; ++j) {
            int idx = fr.ReadInt("Error reading var idx for Clique");
            if (idx < 0 || idx >= varCount) throw Error(GR_E_INVALID, "Invalid var idx " + std::to_string(idx) + " for Clique " + std::to_string(i));
            fv[j] = idx;
        }
        m.Funcs.push_back(m.NewFunction(i, fv));
    }
    for (auto &f : m.Funcs) {
        int ts = fr.ReadInt("Error reading table size");
        if (ts != (int)f.Table.size())
            throw Error(GR_E_INVALID, "Read table size " + std::to_string(ts) + " != previous Clique size " + std::to_string(f.Table.size()) + " on function " + f.Name);
        for (int t = 0; t < ts; ++t) f.Table[t] = fr.ReadFloat("Error reading table entry");
    }
    return m;
}

void UAIReader::ApplyEvidence(const std::string &data, Model &m) {
    int lines = 0;
    std::string text = Preprocess(data, "", &lines);
    if (lines < 1) throw Error(GR_E_INVALID, "Invalid data buffer: there is no data");
    if (lines > 2) throw Error(GR_E_INVALID, "Found " + std::to_string(lines) + " lines: only understand evidence files with 1 or 2 lines");
    FieldReader fr(text);
    if (fr.fields.empty()) throw Error(GR_E_INVALID, "Invalid data: found no fields");
    if (lines == 2) {
        int sc = fr.ReadInt("Error reading UAI evid file sample count");
        if (sc == 0) return;
        if (sc > 1) throw Error(GR_E_INVALID, "Sample count is " + std::to_string(sc) + " - only single sample evidence currently supported");
    }
    int cnt = fr.ReadInt("Error reading UAI evid Variable Count");
    if (cnt < 1) return;
    for (int i = 0; i < cnt; ++i) {
        int idx = fr.ReadInt("Could not read evid var");
        if (idx < 0 || idx >= (int)m.Vars.size()) throw Error(GR_E_INVALID, "Read incorrect variable index " + std::to_string(idx));
        Variable &v = m.Vars[idx];
        if (v.FixedVal != -1) throw Error(GR_E_INVALID, "variable[" + std::to_string(idx) + "]:" + v.Name + " had previous fixedval " + std::to_string(v.FixedVal));
        int val = fr.ReadInt("Could not read evid var value");
        if (val < 0 || val >= v.Card) throw Error(GR_E_INVALID, "Read invalid value " + std::to_string(val) + " for variable[" + std::to_string(idx) + "] with card " + std::to_string(v.Card));
        v.FixedVal = val;
    }
}

std::vector<Variable> UAIReader::ReadMargSolution(const std::string &data) {
    if (data.size() < 11) throw Error(GR_E_INVALID, "Invalid data buffer: len=" + std::to_string(data.size()) + " (<11)");
    int lines = 0;
    std::string text = Preprocess(data, "MAR", &lines); // skips e.g. Merlin's PR preamble
    if (lines < 1) throw Error(GR_E_INVALID, "No lines in file");
    FieldReader fr(text);
    if (fr.fields.size() < 4) throw Error(GR_E_INVALID, "Invalid data: only " + std::to_string(fr.fields.size()) + " fields found (<4)");
    std::string typ = fr.Read("Could not understand file");
    if (typ != "MAR") throw Error(GR_E_INVALID, "Unknown solution file type " + typ);
    int varCount = fr.ReadCount("Error reading UAI MAR Solution Variable Count");
    if (varCount < 1) throw Error(GR_E_INVALID, "Invalid variable count: " + std::to_string(varCount));
    std::vector<Variable> out;
    out.reserve(varCount);
    for (int i = 0; i < varCount; ++i) {
        int card = fr.ReadCount("Error reading Card for var");
        if (card < 1) throw Error(GR_E_INVALID, "Invalid card " + std::to_string(card) + " for var " + std::to_string(i));
        Variable v = Variable::New(i, card);
        for (int k = 0; k < card; ++k) {
            double p = fr.ReadFloat("Could not read marg prob");
            if (p < 0.0 || p > 1.0) throw Error(GR_E_INVALID, "Invalid marg prob on var " + std::to_string(i));
            v.Marginal[k] = p;
        }
        v.NormMarginal();
        out.push_back(std::move(v));
    }
    return out;
}

std::string ReadFile(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(GR_E_INVALID, "Could not READ " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

Model NewModelFromBuffer(const std::string &data) {
    Model m = UAIReader::ReadModel(data);
    m.Check();
    return m;
}

Model NewModelFromFile(const std::string &filename, bool useEvidence) {
    Model m = NewModelFromBuffer(ReadFile(filename));
    size_t dot = filename.find_last_of('.');
    size_t slash = filename.find_last_of('/');
    m.Name = (dot != std::string::npos && (slash == std::string::npos || dot > slash)) ? filename.substr(0, dot) : filename;
    if (useEvidence) {
        for (auto &v : m.Vars) v.FixedVal = -1;
        UAIReader::ApplyEvidence(ReadFile(filename + ".evid"), m);
    }
    return m;
}

std::vector<Variable> NewSolutionFromFile(const std::string &filename) { return UAIReader::ReadMargSolution(ReadFile(filename)); }

// ------------------------------------------------------------------ metrics
static void totals(const double *a, const double *b, int card, double &t1, double &t2) {
    const double eps = 1e-12;
    t1 = t2 = 0;
    for (int c = 0; c < card; ++c) { t1 += a[c]; t2 += b[c]; }
    if (t1 < eps) t1 = eps;
    if (t2 < eps) t2 = eps;
}

double MaxAbsDiff(const double *a, const double *b, int card) {
    double t1, t2, mx = 0;
    totals(a, b, card, t1, t2);
    for (int c = 0; c < card; ++c) {
        double e = std::fabs(a[c] / t1 - b[c] / t2);
        if (c == 0 || e > mx) mx = e;
    }
    return mx;
}

double MeanAbsDiff(const double *a, const double *b, int card) {
    if (card < 1) return 0;
    double t1, t2, s = 0;
    totals(a, b, card, t1, t2);
    for (int c = 0; c < card; ++c) s += std::fabs(a[c] / t1 - b[c] / t2);
    return s / (double)card;
}

double HellingerDiff(const double *a, const double *b, int card) {
    double t1, t2, s = 0;
    totals(a, b, card, t1, t2);
    for (int c = 0; c < card; ++c) {
        double d = std::sqrt(a[c] / t1) - std::sqrt(b[c] / t2);
        s += d * d;
    }
    return std::sqrt(s) / std::sqrt(2.0);
}

static double klDivergence(const std::vector<double> &p, const std::vector<double> &q) {
    const double eps = 1e-12;
    double d = 0;
    for (size_t i = 0; i < p.size(); ++i) {
        double p1 = std::max(p[i], eps), p2 = std::max(q[i], eps);
        d += p1 * std::log2(p1 / p2);
    }
    return d;
}

double JSDivergence(const double *a, const double *b, int card) {
    double t1, t2;
    totals(a, b, card, t1, t2);
    std::vector<double> p1(card), p2(card), mid(card);
    for (int i = 0; i < card; ++i) {
        p1[i] = a[i] / t1;
        p2[i] = b[i] / t2;
        mid[i] = (p1[i] + p2[i]) * 0.5;
    }
    return 0.5 * (klDivergence(p1, mid) + klDivergence(p2, mid));
}

ErrorSuite NewErrorSuite(int nVars, const int32_t *card, const int32_t *fixed1, const int32_t *fixed2, const double *m1,
                         const double *m2) {
    int count = 0;
    for (int i = 0; i < nVars; ++i) {
        int f1 = fixed1 ? fixed1[i] : -1, f2 = fixed2 ? fixed2[i] : -1;
        if (f1 < 0 && f2 < 0) count++;
    }
    if (count < 1) throw Error(GR_E_INVALID, "No un-fixed vars to score");
    ErrorSuite es;
    int off = 0;
    for (int i = 0; i < nVars; ++i) {
        const int c = card[i];
        const bool fixed = (fixed1 && fixed1[i] >= 0) || (fixed2 && fixed2[i] >= 0); // fixed vars score 0
        double d;
        d = fixed ? 0.0 : MeanAbsDiff(m1 + off, m2 + off, c);
        es.MeanMeanAbsError += d; es.MaxMeanAbsError = std::max(d, es.MaxMeanAbsError);
        d = fixed ? 0.0 : MaxAbsDiff(m1 + off, m2 + off, c);
        es.MeanMaxAbsError += d; es.MaxMaxAbsError = std::max(d, es.MaxMaxAbsError);
        d = fixed ? 0.0 : HellingerDiff(m1 + off, m2 + off, c);
        es.MeanHellinger += d; es.MaxHellinger = std::max(d, es.MaxHellinger);
        d = fixed ? 0.0 : JSDivergence(m1 + off, m2 + off, c);
        es.MeanJSDiverge += d; es.MaxJSDiverge = std::max(d, es.MaxJSDiverge);
        off += c;
    }
    es.MeanMeanAbsError /= count;
    es.MeanMaxAbsError /= count;
    es.MeanHellinger /= count;
    es.MeanJSDiverge /= count;
    return es;
}

} // namespace gr
