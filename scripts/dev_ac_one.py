import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import grample_amd as g
m = g.Model.NewModelFromFile(os.path.join(ROOT, "tests", "golden", "res", "Grids_13.uai"))
s = g.HIPChains(m, 65536, seed=42)
s.BurnIn(1000)
s.AdvanceChain(1000)
s.AdvanceChain(1000)
