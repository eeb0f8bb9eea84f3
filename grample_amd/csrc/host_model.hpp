// Host-side model types: the C++ mirror of grample's `model` package for the
// pieces the hot path touches (Variable, Function, Model, UAIReader, Solution,
// ErrorSuite).  Same names, argument meaning and error behaviour as the Go
// code so a host written against model/*.go maps one to one.  Errors are
// reported as gr::Error (message text follows the reference's wording).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace gr {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// model/variable.go:10-18
struct Variable {
    int ID = 0;
    std::string Name;
    int Card = 0;
    int FixedVal = -1;
    std::vector<double> Marginal;
    bool Collapsed = false;

    static Variable New(int index, int card); // NewVariable (variable.go:22-51): uniform marginal
    void NormMarginal();                      // variable.go:106-147
    void Check() const;                       // variable.go:76-103
};

// model/function.go:37-42.  Vars holds variable IDs (most significant first).
struct Function {
    std::string Name;
    std::vector<int> Vars;
    std::vector<double> Table;
    bool IsLog = false;
};

constexpr int kMaxTabSize = 1 << 23; // function.go:59

// model/model.go:24-29
struct Model {
    std::string Type;
    std::string Name;
    std::vector<Variable> Vars;
    std::vector<Function> Funcs;

    int TotalCard() const;
    int CalcTabSize(const std::vector<int> &vars) const;          // function.go:47-57
    Function NewFunction(int index, const std::vector<int> &vars) const; // function.go:62-93
    int CalcIndex(const Function &f, const int *values, int n) const;    // function.go:180-202 (throws)
    double Eval(const Function &f, const int *values, int n) const;      // function.go:146-157 (throws)
    static void UseLogSpace(Function &f);                                // function.go:126-142
    void Check() const;                                                  // model.go:115-157
};

// model/uai.go
struct UAIReader {
    static std::string Preprocess(const std::string &data, const std::string &reqPrefix, int *lineCount); // uai.go:20-50
    static Model ReadModel(const std::string &data);                  // uai.go:53-179
    static void ApplyEvidence(const std::string &data, Model &m);     // uai.go:183-249
    static std::vector<Variable> ReadMargSolution(const std::string &data); // uai.go:252-332
};

Model NewModelFromBuffer(const std::string &data);                    // model.go:79-91
Model NewModelFromFile(const std::string &filename, bool useEvidence);// model.go:52-76
std::vector<Variable> NewSolutionFromFile(const std::string &filename); // solution.go:21-33

// model/error.go:15-78
struct ErrorSuite {
    double MeanMeanAbsError = 0, MeanMaxAbsError = 0, MeanHellinger = 0, MeanJSDiverge = 0;
    double MaxMeanAbsError = 0, MaxMaxAbsError = 0, MaxHellinger = 0, MaxJSDiverge = 0;
};
double MaxAbsDiff(const double *a, const double *b, int card);   // error.go:81-113
double MeanAbsDiff(const double *a, const double *b, int card);  // error.go:116-151
double HellingerDiff(const double *a, const double *b, int card);// error.go:158-190
double JSDivergence(const double *a, const double *b, int card); // error.go:217-249
ErrorSuite NewErrorSuite(int nVars, const int32_t *card, const int32_t *fixed1, const int32_t *fixed2,
                         const double *m1, const double *m2);    // error.go:28-78

std::string ReadFile(const std::string &path);

} // namespace gr
