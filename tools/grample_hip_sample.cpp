// grample_hip_sample -- C++ host driver above the C ABI (include/grample_hip.h).
//
// Mirrors what `grample sample` (cmd/root.go:309-585) does around the hot path, with the same flag
// names: read model (+ evidence, + .MAR solution), build chains, burn-in, then loop
// advance -> MergeChains -> ErrorSuite status line, final report.  Everything below the loop is
// the HIP library; this file contains no sampling code.  The reference's Go CLI stays the real
// front end; this is the compiled-language host the image can build (no Go here).
//
//   grample_hip_sample -m MODEL.uai [-s simple|collapsed|adaptive] [-o] [-d] [-c CHAINS] [-k GROUPS]
//                      [-b BURNIN] [-i MAXITERS] [-r ROUNDS] [-w CWIN] [-e SEED] [--device N]
//   -i MAXITERS: like the reference's --maxiters it bounds the TOTAL sample count over all chains (root.go:482-496: the loop
//      ends once sampleCount > maxIters); a round advances every chain by ceil(MAXITERS / (ROUNDS * CHAINS)) updates
//   -w CWIN: every round is one Chain.AdvanceChain with ConvergenceWindow CWIN (chain.go:180-218) instead
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/grample_hip.h"

static void die(const char *what) {
    std::fprintf(stderr, "error: %s: %s\n", what, gr_last_error());
    std::exit(1);
}

int main(int argc, char **argv) {
    std::string model, samp = "simple";
    bool use_evidence = false, use_solution = false;
    long chains = 65536, groups = 0, burnin = -1, maxiters = -1, rounds = 10, seed = 42, cwin = 0;
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "-m" || a == "--model") model = next();
        else if (a == "-s" || a == "--sampler") samp = next();
        else if (a == "-d" || a == "--evidence") use_evidence = true;
        else if (a == "-o" || a == "--solution") use_solution = true;
        else if (a == "-c" || a == "--chains") chains = std::atol(next());
        else if (a == "-k" || a == "--groups") groups = std::atol(next());
        else if (a == "-b" || a == "--burnin") burnin = std::atol(next());
        else if (a == "-i" || a == "--maxiters") maxiters = std::atol(next());
        else if (a == "-r" || a == "--rounds") rounds = std::atol(next());
        else if (a == "-w" || a == "--cwin") cwin = std::atol(next());
        else if (a == "-e" || a == "--seed") seed = std::atol(next());
        else if (a == "--device") device = std::atoi(next());
        else { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
    }
    if (model.empty()) { std::fprintf(stderr, "usage: %s -m MODEL.uai [-s simple|collapsed] [-o] [-d] [-c chains] ...\n", argv[0]); return 2; }
    const bool collapsed = samp == "collapsed", adaptive = samp == "adaptive";
    if (!collapsed && !adaptive && samp != "simple") { std::fprintf(stderr, "Unknown Sampler: %s\n", samp.c_str()); return 2; }

    std::printf("Reading model from %s\n", model.c_str());
    gr_model *m = gr_model_from_file(model.c_str(), use_evidence ? 1 : 0);
    if (!m) die("Could not READ model");
    const int n = gr_model_n_vars(m), H = gr_model_total_card(m);
    std::printf("Model has %d vars and %d functions\n", n, gr_model_n_funcs(m));
    std::vector<int32_t> card(n), fixed(n);
    gr_model_cards(m, card.data());
    gr_model_fixed(m, fixed.data());

    std::vector<double> sol;
    if (use_solution) {
        gr_solution *s = gr_solution_from_file((model + ".MAR").c_str());
        if (!s) die("Could not read solution file");
        sol.resize(H);
        gr_solution_marginals(s, sol.data());
        gr_solution_destroy(s);
    }
    // defaults as in cmd/root.go:348-356
    if (burnin < 0) burnin = 2000L * n;
    if (maxiters < 0) maxiters = 20000L * n;
    if (groups <= 0) groups = (collapsed || adaptive) ? 128 : 1; // reference MaxChains (adaptive.go:49)
    if (chains % groups) chains -= chains % groups;

    gr_sampler_opts o{};
    o.struct_size = sizeof o;
    o.n_chains = (uint32_t)chains;
    o.n_variants = (uint32_t)groups;
    o.device = device;
    o.seed = (uint64_t)seed;
    gr_sampler *s = gr_sampler_create(m, &o);
    if (!s) die("Could not create sampler");
    std::printf("Creating %ld chains in %ld group(s) and performing burn-in (%ld)\n", chains, groups, burnin);
    if (collapsed)
        for (long k = 0; k < groups; ++k) {
            int32_t cv = -1;
            std::vector<double> marg(64);
            if (gr_collapse(s, (uint32_t)k, -1, &cv, marg.data())) die("Could not collapse random var on startup");
            if (k < 4) std::printf("        - group %ld collapsed variable %d\n", k, cv);
        }
    if (adaptive && gr_set_active_groups(s, 2)) die("Could not set the base chain count"); // root.go:360-363: at least 2 base chains
    if (gr_burnin(s, (uint64_t)burnin)) die("Failure during chain burn in");

    std::printf("Main Sampling Start\n");
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<double> merged(H);
    std::vector<int32_t> coll(n);
    // --maxiters counts samples over ALL chains (root.go:486-496), so one round gives every chain its share of it
    const uint64_t per_round = (uint64_t)std::max(1L, (maxiters + rounds * chains - 1) / std::max(1L, rounds * chains));
    double es[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool scored = false;
    for (long r = 0; r < rounds; ++r) {
        if (cwin > 0) {
            uint64_t ran = 0;
            if (gr_advance_chain(s, (uint32_t)cwin, &ran)) die("Async sample generation failed");
            if (r == 0) std::printf("  AdvanceChain(cw=%ld) ran %llu updates per chain\n", cwin, (unsigned long long)ran);
        } else if (adaptive ? gr_advance_window(s, per_round) : gr_advance(s, per_round)) die("Async sample generation failed");
        uint64_t samples = 0;
        gr_total_samples(s, &samples);
        gr_sampler *one[1] = {s};
        if (gr_merge(one, 1, merged.data(), coll.data())) die("Could not merge chains to calculate score");
        const double rt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("  Samps: %14llu | RT %10.3fsec | %.3e updates/s", (unsigned long long)samples, rt, samples / rt);
        if (use_solution) {
            if (gr_error_suite(n, card.data(), nullptr, fixed.data(), sol.data(), merged.data(), es)) die("Error calculating score");
            std::printf(" | MeanHellinger %.6f MaxHellinger %.6f MeanMAE %.6f", es[2], es[6], es[0]);
            scored = true;
        }
        std::printf("\n");
        if (cwin <= 0 && samples > (uint64_t)maxiters) break; // root.go:493-496: sampleCount > maxIters ends the run
        if (adaptive && r < rounds / 2) { // adaptation only during the first half of the run (root.go:469,542-547)
            std::vector<int32_t> nv(4);
            uint32_t nn = 0;
            const int pre = gr_active_groups(s);
            if (gr_adapt(s, 4, nv.data(), &nn, nullptr)) die("Adapt failed");
            if (nn) std::printf("ADAPT: %d Chain groups (was %d)\n", gr_active_groups(s), pre);
        }
    }
    std::printf("DONE\n");
    if (use_solution && scored) std::printf("FINAL MeanHellinger=%.8f MaxHellinger=%.8f MeanJS=%.8f MaxJS=%.8f\n", es[2], es[6], es[3], es[7]);
    gr_sampler_destroy(s);
    gr_model_destroy(m);
    return 0;
}
