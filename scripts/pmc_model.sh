#!/bin/bash
# SQ counter passes over one launch of a fixture model: bash scripts/pmc_model.sh MODEL.uai CHAINS FORM SWEEPS
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=$1; N=$2; F=$3; S=$4
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_model -o run -- python3 $R/scripts/dev_od_one.py $N $F $S $M > $R/gpurun_out/pmc_model.log 2>&1 || exit 1
tail -1 $R/gpurun_out/pmc_model.log
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(list)
for f in glob.glob(f"{R}/gpurun_out/pmc_model/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sweep" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k[0], f"{k[1]:20s} last={v[-1]:.5g}")
PY
