# One gpurun call that produces every file of profiles/ for the current round (copy gpurun_out/r6/* to profiles/round6_* afterwards:
# tools/collect_profiles.sh).  Before the call, HERE (hipcc cross-compiles): rebuild the experiment libraries the probes load --
#   python tools/variant_build.py pptl -DDPN_TIMELINE -DPP_TIMELINE               (tools/_variants/libdpn_hip_pptl.so: pp_timeline.py)
#   python tools/variant_build.py tl -DDPN_TIMELINE -DTS_TIMELINE                 (tools/_variants/libdpn_hip_tl.so: tiles_timeline.py, bwd_tiles_timeline.py)
#   python -m deepphysinet_amd.build --experiments                                (libdpn_hip_exp.so: the shelved kernels' own tests)
# (Round 3's / 4's micro-benchmarks, hand-scheduled k-step variants, power / clock tables and encoder timelines concern code this round did not
# touch: their round3_* / round4_* files stand.)
set -x
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
R=gpurun_out/r6; mkdir -p $R
export MASTER_ADDR=127.0.0.1
timeout 2700 python -m pytest tests -q -m gpu > $R/tests_gpu.txt 2>&1; tail -4 $R/tests_gpu.txt
timeout 900 python bench.py > $R/bench_bf16x2.json 2> $R/bench_bf16x2.err
timeout 900 python bench.py --steps 20 --warmup 5 > $R/bench_bf16x2_driver_protocol.json 2>> $R/bench_bf16x2.err
timeout 900 python bench.py --prec bf16 --no-cpu-baseline > $R/bench_bf16.json 2> $R/bench_bf16.err
timeout 600 python bench.py --leads 61 --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-power > $R/bench_cfg2_61leads_bf16x2.json 2> $R/bench_cfg2.err
timeout 600 python bench.py --leads 61 --steps 10 --warmup 2 --prec bf16 --no-cpu-baseline --no-alt --no-power > $R/bench_cfg2_61leads_bf16.json 2>> $R/bench_cfg2.err
timeout 600 python bench.py --encoder-fp8 --no-cpu-baseline --no-alt --no-power --no-lead-probe > $R/bench_cfg4_encoder_fp8_mx.json 2> $R/bench_cfg4.err
DPN_BENCH_RCCL_ONE_RANK=1 DPN_BENCH_CAPTURE_COLLECTIVES=0 MASTER_PORT=29581 timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-alt --no-power --no-lead-probe 2> $R/bench_rccl1.err | grep '^{' > $R/bench_bf16x2_rccl_one_rank.json
DPN_BENCH_RCCL_ONE_RANK=1 DPN_BENCH_CAPTURE_COLLECTIVES=1 MASTER_PORT=29582 timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-alt --no-power --no-lead-probe 2>> $R/bench_rccl1.err | grep '^{' > $R/bench_bf16x2_rccl_one_rank_one_graph.json
DPN_BENCH_RCCL_ONE_RANK=1 DPN_BENCH_TRY_FORMS=1 MASTER_PORT=29583 timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-alt --no-power --no-lead-probe 2>> $R/bench_rccl1.err | grep '^{' > $R/bench_bf16x2_rccl_one_rank_form_trial.json
DPN_BENCH_RCCL_ONE_RANK=1 DPN_BENCH_TRY_FORMS=1 DPN_BENCH_TRIAL_TEST_STALL=replay DPN_BENCH_TRIAL_WATCHDOG_S=5 MASTER_PORT=29584 timeout 600 python bench.py --steps 50 --no-cpu-baseline --no-alt --no-power --no-lead-probe 2>> $R/bench_rccl1.err | grep '^{' > $R/bench_bf16x2_rccl_one_rank_trial_stall_fallback.json
DPN_BENCH_ONE_DEVICE=1 DPN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-power --no-lead-probe 2> $R/bench_2ranks.err | grep '^{' > $R/bench_2ranks_one_device_gloo.json
DPN_BENCH_ONE_DEVICE=1 DPN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --leads 3 --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-power 2>> $R/bench_2ranks.err | grep '^{' > $R/bench_2ranks_one_device_gloo_3leads.json
DPN_BENCH_ONE_DEVICE=1 DPN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --points 4096 --steps 10 --warmup 2 --blocks 3 --no-cpu-baseline --no-alt --no-power --no-lead-probe 2>> $R/bench_2ranks.err | grep '^{' > $R/bench_8ranks_one_device_gloo_4096pts.json
timeout 600 python tools/tiles_timeline.py 37265 tl > $R/fwd_tiles_kernel_timeline.txt 2>&1
timeout 300 python tools/bwd_tiles_timeline.py 37265 tl > $R/bwd_tiles_timeline_full.txt 2>&1
timeout 300 python tools/pp_ab.py 37265 5197 1037 129 1 > $R/fwd_pp_vs_tiles.txt 2>&1
timeout 300 python tools/pp_timeline.py 37265 pptl > $R/fwd_pp_kernel_timeline.txt 2>&1
( timeout 1200 python tools/soak.py bf16x2 200; timeout 900 python tools/soak.py bf16 200 ) 2>&1 | grep -v "amdgpu.ids" > $R/soak_bitwise.txt || true
timeout 600 python tools/enc_batch_check.py 1 3 > $R/encoder_vs_fp64.txt 2>&1
rm -rf $R/prof_cfg2; timeout 900 rocprofv3 --kernel-trace --stats -d $R/prof_cfg2 -o trace -- python3 bench.py --leads 61 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-power > $R/bench_prof_cfg2.log 2>&1
timeout 600 python tools/prof_summary.py $(find $R/prof_cfg2 -name "*.db" | head -1) 40 > $R/cfg2_61leads_kernel_stats_bf16x2.txt
for prec in bf16x2 bf16; do
  timeout 600 python tools/phase_times.py $prec > $R/phase_times_$prec.txt 2>&1
  timeout 600 python tools/reference_step.py $prec > $R/reference_shaped_step_$prec.json 2>> $R/refstep.err
  rm -rf $R/prof_$prec; timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof_$prec -o trace -- python3 bench.py --steps 10 --warmup 3 --prec $prec --no-cpu-baseline --no-alt --no-power --no-lead-probe > $R/bench_prof_$prec.log 2>&1
  DB=$(find $R/prof_$prec -name "*.db" | head -1)
  timeout 600 python tools/prof_summary.py $DB 40 > $R/kernel_trace_stats_bench_$prec.txt
  timeout 600 python tools/timeline.py $DB 2 > $R/step_timeline_$prec.txt
  rm -f $R/pmc_eager_step_$prec.txt
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
