"""One sampler, one form (argv[1] = fast_form), Grids_13 @ 65 536 chains, 2 x 100 000 counted updates per chain: the run scripts/dev_bits.sh profiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
m = g.Model.NewModelFromFile(os.path.join(ROOT, "tests", "golden", "res", "Grids_13.uai"))
s = g.HIPChains(m, 65536, seed=7, fast_form=int(sys.argv[1]))
s.BurnIn(1000); s.Advance(100000); s.Advance(100000)
i = s.info()
print(f"fast_form {sys.argv[1]} path {i['path']}: {65536 * 100000 / (i['kernel_ms_last'] * 1e-3):.3e} upd/s (kernel ms {i['kernel_ms_last']:.2f})")
