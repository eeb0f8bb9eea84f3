// Host-code sanitizer harness: parses every fixture, applies evidence, reads solutions, converts to
// log space and compiles the device image (programs, view tables, memoised programs) under
// AddressSanitizer + UBSan.  Built and run by tests/test_host_model.py::test_host_code_under_sanitizers.
#include <cstdio>
#include <string>
#include <vector>

#include "../grample_amd/csrc/compiled_model.hpp"
#include "../grample_amd/csrc/host_model.hpp"

int main(int argc, char **argv) {
    int bad = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string path = argv[i];
        try {
            for (int ev = 0; ev < 2; ++ev) {
                gr::Model m = gr::NewModelFromFile(path, ev != 0);
                for (auto &f : m.Funcs) gr::Model::UseLogSpace(f);
                const gr::VariantImage im = gr::CompileVariant(m, gr::CompileLimits{});
                size_t words = im.prog.size() + im.cprog.size() + im.pprog.size() + im.views.size();
                std::printf("%s ev=%d vars=%zu funcs=%zu free=%zu image_words=%zu cpt=%d pairwise=%d\n", path.c_str(), ev, m.Vars.size(),
                            m.Funcs.size(), im.free_vars.size(), words, (int)im.cpt_ok, (int)im.pairwise);
            }
            try {
                auto sol = gr::NewSolutionFromFile(path + ".MAR");
                std::printf("  solution vars=%zu\n", sol.size());
            } catch (const gr::Error &) {
            }
        } catch (const gr::Error &e) {
            std::printf("ERROR %s: %s\n", path.c_str(), e.what());
            bad = 1;
        }
    }
    // malformed inputs must throw, not crash
    for (const char *txt : {"", "MARKOV", "MARKOV\n1\n2\n1\n1 0\n2\n0.5", "MARKOV\n2\n2 2\n1\n2 0 5\n4\n1 1 1 1\n", "MARKOV\n1\n2\n1\n1 0\n2\n x y\n"}) {
        try {
            gr::NewModelFromBuffer(txt);
            std::printf("ERROR: accepted malformed input\n");
            bad = 1;
        } catch (const gr::Error &) {
        }
    }
    return bad;
}
