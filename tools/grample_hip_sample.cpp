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
//   --experiment -t TRACE: experiment mode (root.go:455-458, 520-533, 586-600): the trace file gets the header and one line per advance
//      round -- RunSecs, MaxHell, NegLogMaxHell, MaxJS, NegLogMaxJS, CollapseCount -- a last line after the final merge and the FINAL
//      STATUS block; needs -o (the scores come from the .MAR solution) and a trace file (root.go:315-317)
//   -x MAXSECS: wall-clock bound of the main loop (root.go:464, 483-485; 0 = none, the rounds / maxiters bounds remain); in adaptive
//      mode Adapt runs only during the first MAXSECS / 2 seconds (root.go:469, 542-547)
//   -a CHAINADDS: chain groups added per Adapt (root.go:202; must be 1 unless the sampler is adaptive, root.go:440-442)
//   --adapt-rounds K (not a reference flag): adapt after each of the first K rounds instead of by wall time -- reproducible runs
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
    long chains = 65536, groups = 0, burnin = -1, maxiters = -1, rounds = 10, seed = 42, cwin = 0, maxsecs = 0, chainadds = -1, adapt_rounds = -1;
    int device = 0;
    bool experiment = false, rounds_given = false;
    std::string trace_file, scan = "random";
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
        else if (a == "-r" || a == "--rounds") { rounds = std::atol(next()); rounds_given = true; }
        else if (a == "-x" || a == "--maxsecs") maxsecs = std::atol(next());
        else if (a == "-a" || a == "--chainadds") chainadds = std::atol(next());
        else if (a == "-t" || a == "--trace") trace_file = next();
        else if (a == "--experiment") experiment = true;
        else if (a == "--adapt-rounds") adapt_rounds = std::atol(next());
        else if (a == "-w" || a == "--cwin") cwin = std::atol(next());
        else if (a == "-e" || a == "--seed") seed = std::atol(next());
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--scan") scan = next(); // random (default: one variable per 64-chain group and update) | strict (the reference's: one per chain) | systematic
        else { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
    }
    if (model.empty()) { std::fprintf(stderr, "usage: %s -m MODEL.uai [-s simple|collapsed|adaptive] [-o] [-d] [-c chains] [--scan random|strict|systematic] ...\n", argv[0]); return 2; }
    const bool collapsed = samp == "collapsed", adaptive = samp == "adaptive";
    if (!collapsed && !adaptive && samp != "simple") { std::fprintf(stderr, "Unknown Sampler: %s\n", samp.c_str()); return 2; }
    if (experiment && trace_file.empty()) { std::fprintf(stderr, "Experiment mode requires a trace file\n"); return 2; }            // root.go:315-317
    if (experiment && !use_solution) { std::fprintf(stderr, "Experiment mode scores against the solution file: add -o\n"); return 2; }
    if (chainadds < 0) chainadds = adaptive ? 4 : 1; // (the reference's default is 1; script/experiment runs with 4)
    if (!adaptive && chainadds != 1) { std::fprintf(stderr, "Sampler is not adaptive: ChainAdds=%ld makes no sense\n", chainadds); return 2; } // root.go:440-442
    if (maxsecs > 0 && !rounds_given) rounds = 1L << 40; // bounded by time (and maxiters) like the reference's loop
    FILE *trace = nullptr;
    if (!trace_file.empty()) {
        trace = std::fopen(trace_file.c_str(), "w"); // root.go:82-88
        if (!trace) { std::fprintf(stderr, "Could not create trace file %s\n", trace_file.c_str()); return 2; }
    }

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
    if (scan == "strict") o.flags |= GR_FLAG_SCAN_PER_CHAIN;
    else if (scan == "systematic") o.flags |= GR_FLAG_SCAN_SYSTEMATIC;
    else if (scan != "random") { std::fprintf(stderr, "unknown --scan %s (random | strict | systematic)\n", scan.c_str()); return 2; }
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

    const auto t0 = std::chrono::steady_clock::now(); // (the reference's startTime is taken before the chains are built; RunSecs here starts with the main loop)
    if (experiment) { // root.go:455-458
        std::fprintf(trace, "// EXPERIMENT RESULTS\n");
        std::fprintf(trace, "RunSecs, MaxHell, NegLogMaxHell, MaxJS, NegLogMaxJS, CollapseCount\n");
    }
    std::printf("Main Sampling Start\n");
    std::vector<double> merged(H);
    std::vector<int32_t> coll(n);
    // --maxiters counts samples over ALL chains (root.go:486-496), so one round gives every chain its share of it
    const uint64_t per_round = (uint64_t)std::max(1L, (maxiters + rounds * chains - 1) / std::max(1L, rounds * chains));
    double es[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool scored = false, keep_adapting = true;
    auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    auto trace_line = [&](double run_time) { // root.go:520-533 / 586-597
        int col = 0;
        for (int v = 0; v < n; ++v) col += coll[v] ? 1 : 0;
        std::fprintf(trace, "%.1f, %.8f, %.5f, %.8f, %.5f, %d\n", run_time, es[6], -std::log2(es[6]), es[7], -std::log2(es[7]), col);
        return col;
    };
    auto error_report = [&](FILE *f, const char *prefix) { // errorReport(.., short = false, ..) (root.go:256-305)
        std::fprintf(f, "%s ... M:mean(neg log), X:max(neg log)\n", prefix);
        const char *titles[4] = {"MeanAbsError", "MaxAbsError", "Hellinger", "JS Diverge"};
        for (int k = 0; k < 4; ++k)
            std::fprintf(f, "%15s => M:%.6f(%7.3f) X:%.6f(%7.3f)\n", titles[k], es[k], -std::log2(es[k]), es[4 + k], -std::log2(es[4 + k]));
    };
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
        if (experiment) trace_line(rt);
        bool keep_working = true;
        if (maxsecs > 0 && rt > (double)maxsecs) keep_working = false;                                  // root.go:483-485
        if ((cwin <= 0 || experiment || maxsecs > 0) && samples > (uint64_t)maxiters) keep_working = false; // root.go:493-496: sampleCount > maxIters
        if (!keep_working) break;
        // Adaptive update (root.go:541-560): while the first half of maxSecs lasts -- or, reproducibly, for the first K rounds
        if (adaptive && keep_adapting) {
            const bool stop = adapt_rounds >= 0 ? r >= adapt_rounds : maxsecs > 0 ? rt > (double)(maxsecs / 2) : r >= rounds / 2;
            if (stop) { std::printf("STOPPING ADAPTATION\n"); keep_adapting = false; }
        }
        if (adaptive && keep_adapting) {
            std::vector<int32_t> nv((size_t)std::max(1L, chainadds));
            uint32_t nn = 0;
            const int pre = gr_active_groups(s);
            if (gr_adapt(s, (uint32_t)chainadds, nv.data(), &nn, nullptr)) die("Adapt failed");
            if (nn) std::printf("ADAPT: %d Chains (was %d)\n", gr_active_groups(s), pre);
        }
    }
    // COMPLETED: final merge and scores (root.go:563-600)
    const double run_time = secs();
    {
        gr_sampler *one[1] = {s};
        if (gr_merge(one, 1, merged.data(), coll.data())) die("Error in final chain merge");
        if (use_solution && gr_error_suite(n, card.data(), nullptr, fixed.data(), sol.data(), merged.data(), es)) die("Error calculating Final Score!");
    }
    std::printf("DONE\n");
    if (use_solution) {
        scored = true;
        error_report(stdout, "FINAL");
        if (experiment) {
            const int col = trace_line(run_time);
            std::fprintf(trace, "// FINAL STATUS\n");
            error_report(trace, "FINAL");
            std::printf("COLLAPSED VARS (%d):", col);
            for (int v = 0; v < n; ++v) if (coll[v]) std::printf(" %d", v);
            std::printf("\n");
        }
    }
    if (trace) {
        // (the reference goes on to dump evidence, every variable and the whole model as JSON, root.go:672-716: the host's own
        // business, nothing the device path produces beyond the merged marginals above)
        std::fprintf(trace, "// OPERATING PARAMS\n");
        std::fprintf(trace, "Model:                  %s\nApply Evidence:         %s\nSolution:               %s\nSampler:                %s\n", model.c_str(),
                     use_evidence ? "true" : "false", use_solution ? "true" : "false", samp.c_str());
        std::fprintf(trace, "Burn In:                %12ld\nConverge Win:           %12ld\nNum Base Chain:         %12ld\nChains Added per Adapt: %12ld\n", burnin, cwin,
                     adaptive ? 2L : groups, chainadds);
        std::fprintf(trace, "Max Iters:              %12ld\nMax Secs:               %12ld\nRnd Seed:               %12ld\nExperiment Mode:        %s\n", maxiters, maxsecs, seed,
                     experiment ? "true" : "false");
        std::fprintf(trace, "Device chains:          %12ld in %ld group(s) of %ld (one group per reference chain)\n", chains, groups, chains / groups);
        std::fclose(trace);
    }
    if (use_solution && scored) std::printf("FINAL MeanHellinger=%.8f MaxHellinger=%.8f MeanJS=%.8f MaxJS=%.8f\n", es[2], es[6], es[3], es[7]);
    gr_sampler_destroy(s);
    gr_model_destroy(m);
    return 0;
}
