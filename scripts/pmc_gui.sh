cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gui -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_gui.err
python3 - <<'PY'
import csv, os
R=os.environ["GRAFT_REPO_ROOT"]
rows=list(csv.DictReader(open(f"{R}/gpurun_out/pmc_gui/run_counter_collection.csv")))
for r in rows:
    if "par4" in r["Kernel_Name"]:
        dur=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
        print(r["Counter_Name"], r["Counter_Value"], "dur_ns", dur, "ratio GHz", float(r["Counter_Value"])/dur)
PY
rocm-smi --showclocks 2>/dev/null | head -20
