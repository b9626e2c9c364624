# One gpurun call that produces every file of profiles/ for the current round (copy gpurun_out/$R/* to profiles/round2_* afterwards).
set -x
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
R=gpurun_out/r2; mkdir -p $R
python bench.py > $R/bench_bf16x2.json 2> $R/bench_bf16x2.err
python bench.py --prec bf16 > $R/bench_bf16.json 2> $R/bench_bf16.err
python bench.py --leads 61 --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $R/bench_cfg2_61leads_bf16x2.json 2> $R/bench_cfg2.err
python bench.py --leads 61 --steps 10 --warmup 2 --prec bf16 --no-cpu-baseline --no-alt > $R/bench_cfg2_61leads_bf16.json 2>> $R/bench_cfg2.err
DPN_BENCH_SPLIT_STEP=1 python bench.py --no-cpu-baseline --no-alt > $R/bench_four_segments_one_gpu.json 2> $R/bench_split.err
DPN_BENCH_ONE_DEVICE=1 DPN_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-alt > $R/bench_2ranks_one_device_gloo.json 2> $R/bench_2ranks.err
for prec in bf16x2 bf16; do
  python tools/phase_times.py $prec > $R/phase_times_$prec.txt 2>&1
  python tools/reference_step.py $prec > $R/reference_shaped_step_$prec.json 2>> $R/refstep.err
  python tools/timeline_probe.py $prec > $R/fwd_kernel_timeline_$prec.txt 2>&1
  rm -rf $R/prof_$prec; rocprofv3 --kernel-trace --stats -d $R/prof_$prec -o trace -- python3 bench.py --steps 10 --warmup 3 --prec $prec --no-cpu-baseline --no-alt > $R/bench_prof_$prec.log 2>&1
  DB=$(find $R/prof_$prec -name "*.db" | head -1)
  python tools/prof_summary.py $DB 30 > $R/kernel_trace_stats_bench_$prec.txt
  python tools/timeline.py $DB 2 > $R/step_timeline_$prec.txt
  for set in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    tag=$(echo $set | cut -d' ' -f1); rm -rf $R/pmc_${prec}_$tag
    rocprofv3 --kernel-trace --pmc $set -d $R/pmc_${prec}_$tag -o pmc -- python3 tools/pmc_run.py $prec > $R/pmc_${prec}_$tag.log 2>&1
    D=$(find $R/pmc_${prec}_$tag -name "*.db" | head -1); python tools/pmc_summary.py $D >> $R/pmc_eager_step_$prec.txt
  done
  python tools/pmc_traffic.py $prec 37265 $(find $R/pmc_${prec}_FETCH_SIZE -name "*.db" | head -1) $(find $R/pmc_${prec}_WRITE_SIZE -name "*.db" | head -1) $R/pmc_traffic.json
done
find $R -name "*.db" -size +1M -delete
ls -la $R
