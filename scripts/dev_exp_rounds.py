"""Per-round kernel time of the adaptive experiment (Segmentation_11, 2 -> 128 groups of 256 chains, AdvanceChain(2000) + Adapt(+4)): a round costs
what ONE 64-chain group costs to walk its ~500 000 steps, whatever the number of groups.  argv[1]: fast_form (0 auto, 3 never wide, 8 / 9 = 8 / 16 lanes)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import grample_amd as g
m = g.Model.NewModelFromFile(os.path.join(ROOT, "tests/golden/res/Segmentation_11.uai"), True)
form = int(sys.argv[1]) if len(sys.argv) > 1 else 0
flags = g.GR_FLAG_SCAN_SYSTEMATIC if len(sys.argv) > 2 and sys.argv[2] == 'sys' else 0  # argv[2] = sys: the systematic scan
s = g.HIPChains(m, 128 * 256, n_variants=128, seed=42, fast_form=form, flags=flags)
s.set_active_groups(2); s.BurnIn(2000)
tot = 0
for r in range(34):
    L = s.AdvanceChain(2000); k = s.info()["kernel_ms_last"]; tot += k
    print(r, "groups", s.active_groups(), "window", L, "kernel ms %.1f" % k, "rate %.3e" % (s.active_groups() * 256 * L / (k * 1e-3)), "path", s.info()["path"], "lanes", s.info()["block_threads"] // 64, "cum %.0f" % tot, flush=True)
    s.Adapt(4)
