#!/bin/bash
# SQ counter passes over one ObjectDetection_13 launch (value-parallel form 2 and chain-per-lane form 1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for form in 2 1; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_od_a$form -o run -- python3 $R/scripts/dev_od_one.py 32768 $form 50 > $R/gpurun_out/pmc_od_a$form.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_od_b$form -o run -- python3 $R/scripts/dev_od_one.py 32768 $form 50 > $R/gpurun_out/pmc_od_b$form.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for d in ("pmc_od_a2", "pmc_od_b2", "pmc_od_a1", "pmc_od_b1"):
    fs = glob.glob(f"{R}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "k_sweep" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(d, k[0], f"{k[1]:24s} n={len(v)} last={v[-1]:.5g}")
PY
