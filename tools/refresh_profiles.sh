set -x
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r1
python bench.py > gpurun_out/r1/bench_bf16.json 2> gpurun_out/r1/bench_bf16.err
python bench.py --prec bf16x2 > gpurun_out/r1/bench_bf16x2.json 2> gpurun_out/r1/bench_bf16x2.err
python bench.py --leads 61 --steps 10 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/r1/bench_cfg2.json 2> gpurun_out/r1/bench_cfg2.err
python tools/phase_times.py > gpurun_out/r1/phase_times.txt 2>&1
python tools/reference_step.py > gpurun_out/r1/refstep_bf16.json 2>gpurun_out/r1/refstep.err
python tools/reference_step.py bf16x2 > gpurun_out/r1/refstep_bf16x2.json 2>>gpurun_out/r1/refstep.err
rm -rf gpurun_out/r1/prof; rocprofv3 --kernel-trace --stats -d gpurun_out/r1/prof -o trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/r1/bench_prof.log 2>&1
DB=$(find gpurun_out/r1/prof -name "*.db" | head -1); echo DB=$DB
python tools/prof_summary.py $DB 30 > gpurun_out/r1/kernel_trace_stats.txt
python tools/timeline.py $DB 2 > gpurun_out/r1/timeline.txt
for set in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  tag=$(echo $set | cut -d' ' -f1); rm -rf gpurun_out/r1/pmc_$tag
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/r1/pmc_$tag -o pmc -- python3 tools/pmc_run.py bf16 > gpurun_out/r1/pmc_$tag.log 2>&1
  D=$(find gpurun_out/r1/pmc_$tag -name "*.db" | head -1); python tools/pmc_summary.py $D >> gpurun_out/r1/pmc_summary.txt
done
find gpurun_out/r1 -name "*.db" -size +20M -delete
ls -la gpurun_out/r1
