"""Per-block run times of ONE k_sweep_par4 launch (Grids_13, 65 536 chains = 4 blocks per CU): GR_DEV_BLOCK_TIMES makes the library write
{block, start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID} of the last launch to a file.  profiles/r03_block_times.txt: before / after wave_prio.h."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import grample_amd as g
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
m = g.Model.NewModelFromFile(os.path.join(ROOT, "tests/golden/res/Grids_13.uai"))
s = g.HIPChains(m, 65536, seed=7, fast_form=2)
s.BurnIn(1000); s.Advance(100000)
os.environ["GR_DEV_BLOCK_TIMES"] = os.path.join(ROOT, "gpurun_out/block_times.txt")
s.Advance(50000)
d = np.loadtxt(os.environ["GR_DEV_BLOCK_TIMES"], dtype=np.uint64)
t0, t1, hw, xcc = d[:, 1].astype(np.int64), d[:, 2].astype(np.int64), d[:, 3], d[:, 4]
base = t0.min()
dur = (t1 - t0) / 100.0  # us (100 MHz)
print("blocks", len(d), "start spread us", (t0.max() - base) / 100.0, "end spread us", (t1.max() - t1.min()) / 100.0)
print("duration us: min %.1f mean %.1f max %.1f std %.1f" % (dur.min(), dur.mean(), dur.max(), dur.std()))
print("kernel span us", (t1.max() - base) / 100.0, " mean busy fraction", dur.mean() / ((t1.max() - base) / 100.0))
x = (xcc & np.uint64(0xf)).astype(int)
for k in sorted(set(x)):
    sel = x == k
    print(f"xcc {k}: blocks {sel.sum():4d} mean dur {dur[sel].mean():8.1f} max {dur[sel].max():8.1f} min {dur[sel].min():8.1f}")
cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(int)
key = x * 1000 + se * 16 + cu
cnt = np.bincount(key)
print("blocks per (xcc,se,cu): distinct CUs", (cnt > 0).sum(), "histogram of blocks/CU", np.bincount(cnt[cnt > 0]))
