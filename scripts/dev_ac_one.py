"""One model, two AdvanceChain windows (for rocprofv3 --kernel-trace --stats): python scripts/dev_ac_one.py MODEL CHAINS CW"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
name = sys.argv[1] if len(sys.argv) > 1 else "Grids_13.uai"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
cw = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
m = g.Model.NewModelFromFile(os.path.join(ROOT, "tests", "golden", "res", name))
s = g.HIPChains(m, nch, seed=42)
s.BurnIn(1000)
L = s.AdvanceChain(cw)
L = s.AdvanceChain(cw)
s.Advance(L)
