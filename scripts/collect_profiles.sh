#!/bin/bash
# Measurement pass on the GPU box (gpurun -- 'bash scripts/collect_profiles.sh r03'): bench line, rocprofv3 kernel stats of the
# same command, and the PMC passes (separate runs, --pmc with --kernel-trace only) that bench.py's roofline object is built from.
# Everything lands under gpurun_out/<tag>/; scripts/store_profiles.py <tag> copies the summaries into profiles/ and writes
# profiles/roofline_inputs.json.  QUICK=1 skips the long side measurements (all models, seed noise, ...).
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
$R/scripts/ubench/issue_rate.bin > $O/issue_rate_ubench.txt 2>&1
rm -f $O/seed_noise.txt
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
echo "bench done"; head -c 400 $O/bench.json; echo
B="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $B > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- $B > /dev/null 2> $O/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- $B > /dev/null 2> $O/pmc_write.err || exit 1
echo "hbm pmc done"
B2="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sqA -o run -- $B2 > /dev/null 2> $O/pmc_sqA.err || exit 1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O/pmc_sqB -o run -- $B2 > /dev/null 2> $O/pmc_sqB.err || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_gui -o run -- $B2 > /dev/null 2> $O/pmc_gui.err || exit 1
echo "sq pmc done"
python3 $R/scripts/isa_mix.py --json > $O/isa_mix.json 2> $O/isa_mix.err
[ -n "$QUICK" ] && { echo "quick pass done"; exit 0; }
python3 $R/scripts/bench_configs.py > $O/configs.jsonl 2> $O/configs.err || exit 1
echo "configs done"
python3 $R/scripts/dev_time.py > $O/dev_time.txt 2>&1
python3 $R/scripts/dev_od.py > $O/dev_od.txt 2>&1
python3 $R/scripts/dev_wide.py > $O/dev_wide.txt 2>&1
python3 $R/scripts/dev_systematic.py > $O/dev_systematic.txt 2>&1
python3 $R/scripts/dev_all_models.py 32768 > $O/all_models_32768.txt 2>&1
python3 $R/scripts/parity_configs.py > $O/parity_configs.jsonl 2> $O/parity_configs.err
for spec in "ObjectDetection_13.uai 256 50 400" "Segmentation_11.uai 256 20 100" "Grids_13.uai 256 20 200"; do
  echo "== $spec (model chains burn-in-sweeps counted-sweeps; 6 seeds each)" >> $O/seed_noise.txt
  python3 $R/scripts/dev_noise.py $spec 2>&1 | tail -8 >> $O/seed_noise.txt
done
python3 $R/scripts/dev_advance_chain.py > $O/advance_chain.txt 2>&1
python3 $R/scripts/dev_chains.py > $O/dev_chains.txt 2>&1
python3 $R/scripts/dev_bits.py > $O/dev_bits.txt 2>&1
python3 $R/scripts/dev_exp_rounds.py 0 > $O/adaptive_rounds.txt 2>&1
python3 $R/scripts/dev_prio.py > $O/dev_prio.txt 2>&1
GR_DEV_PRIO_SHIFT=32 python3 $R/scripts/dev_prio.py >> $O/dev_prio.txt 2>&1
$R/scripts/ubench/state_form.bin > $O/state_form_ubench_run.txt 2>&1
echo "all done"
