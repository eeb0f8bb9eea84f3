import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.dev_time import timing
name = sys.argv[1] if len(sys.argv) > 1 else "Grids_13.uai"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 160
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
for form in [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0").split(",")]:
    timing(name, nch, sweeps, flags, 0, 1, form)
