# One gpurun call that produces every file of profiles/ for the current round (copy gpurun_out/$R/* to profiles/round3_* afterwards).
# Before the call, HERE (hipcc cross-compiles): rebuild the two experiment libraries the probes load, or they miss entry points added since --
#   python tools/timeline_probe.py --build -DDPN_EXPERIMENT_SPLITS      (libdpn_hip_timeline.so: wgrad_overlap_probe.py's range plans)
#   python tools/variant_build.py tl -DDPN_TIMELINE -DTS_TIMELINE       (libdpn_hip_tl.so: tiles_timeline.py)
# and the micro-benchmarks under tools/microbench/ (hipcc --offload-arch=gfx950 -O3 -o X X.hip).
set -x
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
R=gpurun_out/r3; mkdir -p $R
export MASTER_ADDR=127.0.0.1
timeout 600 python bench.py > $R/bench_bf16x2.json 2> $R/bench_bf16x2.err
timeout 600 python bench.py --prec bf16 > $R/bench_bf16.json 2> $R/bench_bf16.err
timeout 600 python bench.py --leads 61 --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $R/bench_cfg2_61leads_bf16x2.json 2> $R/bench_cfg2.err
timeout 600 python bench.py --leads 61 --steps 10 --warmup 2 --prec bf16 --no-cpu-baseline --no-alt > $R/bench_cfg2_61leads_bf16.json 2>> $R/bench_cfg2.err
timeout 600 python bench.py --encoder-fp8 --steps 100 --no-cpu-baseline --no-alt > $R/bench_cfg4_encoder_fp8.json 2> $R/bench_cfg4.err
DPN_BENCH_RCCL_ONE_RANK=1 MASTER_PORT=29581 timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-alt 2> $R/bench_rccl1.err | grep '^{' > $R/bench_bf16x2_rccl_one_rank.json
DPN_BENCH_RCCL_ONE_RANK=1 DPN_BENCH_CAPTURE_COLLECTIVES=1 MASTER_PORT=29582 timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-alt 2>> $R/bench_rccl1.err | grep '^{' > $R/bench_bf16x2_rccl_one_rank_one_graph.json
DPN_BENCH_ONE_DEVICE=1 DPN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-alt 2> $R/bench_2ranks.err | grep '^{' > $R/bench_2ranks_one_device_gloo.json
timeout 600 python tools/fwd_ab.py bf16x2 37265 > $R/fwd_ring_vs_tiles.txt 2>&1
timeout 600 python tools/bwd_ab.py bf16x2 37265 > $R/bwd_ring_vs_tiles.txt 2>&1
timeout 600 python tools/tiles_timeline.py 37265 tl > $R/fwd_tiles_kernel_timeline.txt 2>&1
timeout 600 ./tools/microbench/l2_stream > $R/microbench_l2_stream.txt 2>&1
timeout 600 ./tools/microbench/l2_stream2 > $R/microbench_l2_stream2.txt 2>&1
timeout 600 ./tools/microbench/kstep_loop > $R/microbench_kstep_loop.txt 2>&1
timeout 600 ./tools/microbench/grid_barrier > $R/microbench_grid_barrier_run.txt 2>&1
P2="10,11,10,11 11,11,9,11 10,10,9,13 10,10,8,14 9,9,8,16 10,10,10,12"
timeout 600 python tools/wgrad_overlap_probe.py bf16x2 10,10,9,13 $P2 $P2 > $R/wgrad_plans.txt 2>&1
P1="10,11,10,11 11,11,10,10 10,10,9,13 11,11,9,11"
timeout 600 python tools/wgrad_overlap_probe.py bf16 10,11,10,11 $P1 $P1 >> $R/wgrad_plans.txt 2>&1
for prec in bf16x2 bf16; do
  timeout 600 python tools/phase_times.py $prec > $R/phase_times_$prec.txt 2>&1
  timeout 600 python tools/reference_step.py $prec > $R/reference_shaped_step_$prec.json 2>> $R/refstep.err
  rm -rf $R/prof_$prec; timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof_$prec -o trace -- python3 bench.py --steps 10 --warmup 3 --prec $prec --no-cpu-baseline --no-alt > $R/bench_prof_$prec.log 2>&1
  DB=$(find $R/prof_$prec -name "*.db" | head -1)
  timeout 600 python tools/prof_summary.py $DB 30 > $R/kernel_trace_stats_bench_$prec.txt
  timeout 600 python tools/timeline.py $DB 2 > $R/step_timeline_$prec.txt
  for set in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TA_BUSY_avr TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | cut -d' ' -f1); rm -rf $R/pmc_${prec}_$tag
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/pmc_${prec}_$tag -o pmc -- python3 tools/pmc_run.py $prec > $R/pmc_${prec}_$tag.log 2>&1
    D=$(find $R/pmc_${prec}_$tag -name "*.db" | head -1); python tools/pmc_summary.py $D >> $R/pmc_eager_step_$prec.txt
  done
  timeout 600 python tools/pmc_traffic.py $prec 37265 $(find $R/pmc_${prec}_FETCH_SIZE -name "*.db" | head -1) $(find $R/pmc_${prec}_WRITE_SIZE -name "*.db" | head -1) $R/pmc_traffic.json
done
find $R -name "*.db" -size +1M -delete
find $R -name "*.log" -size +200k -delete
ls -la $R
