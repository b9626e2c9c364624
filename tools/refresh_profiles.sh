# One gpurun call that produces every file of profiles/ for the current round (copy gpurun_out/r4/* to profiles/round4_* afterwards:
# tools/collect_profiles.sh).  Before the call, HERE (hipcc cross-compiles): rebuild the experiment libraries the probes load --
#   python tools/variant_build.py tl -DDPN_TIMELINE -DTS_TIMELINE                 (libdpn_hip_tl.so: tiles_timeline.py, bwd_tiles_timeline.py)
#   python tools/variant_build.py tlnostore|tlnomfma|tlnoaload|tlnosincos -DDPN_TIMELINE -DTS_TIMELINE -DTS_ABL_NOSTORE|...   (ablations of the tile kernels)
#   python tools/microbench/gen_kstep_asm.py && hipcc --offload-arch=gfx950 -O3 -w -o tools/microbench/kstep_asm tools/microbench/kstep_asm.hip
#   python tools/variant_build.py enctl --unit=5 -DDPN_ENC_TIMELINE               (libdpn_hip_enctl.so: enc_timeline.py)
# (Round 3's micro-benchmarks, ring-vs-tile-split A/B runs and weight-gradient range plans concern kernels this round did not touch: their
# round3_* files stand.)
set -x
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
R=gpurun_out/r4; mkdir -p $R
export MASTER_ADDR=127.0.0.1
timeout 900 python bench.py > $R/bench_bf16x2.json 2> $R/bench_bf16x2.err
timeout 900 python bench.py --prec bf16 --no-cpu-baseline > $R/bench_bf16.json 2> $R/bench_bf16.err
DPN_ENCODER_UNFUSED=1 timeout 600 python bench.py --no-cpu-baseline --no-alt --no-power > $R/bench_bf16x2_per_gemm_encoder_of_round3.json 2>> $R/bench_bf16x2.err
timeout 600 python bench.py --leads 61 --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-power > $R/bench_cfg2_61leads_bf16x2.json 2> $R/bench_cfg2.err
DPN_ENCODER_UNFUSED=1 timeout 600 python bench.py --leads 61 --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-power > $R/bench_cfg2_61leads_bf16x2_per_gemm_encoder_NaN_state.json 2>> $R/bench_cfg2.err
timeout 600 python bench.py --leads 61 --steps 10 --warmup 2 --prec bf16 --no-cpu-baseline --no-alt --no-power > $R/bench_cfg2_61leads_bf16.json 2>> $R/bench_cfg2.err
DPN_BENCH_RCCL_ONE_RANK=1 MASTER_PORT=29581 timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-alt --no-power 2> $R/bench_rccl1.err | grep '^{' > $R/bench_bf16x2_rccl_one_rank.json
DPN_BENCH_RCCL_ONE_RANK=1 DPN_BENCH_CAPTURE_COLLECTIVES=1 MASTER_PORT=29582 timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-alt --no-power 2>> $R/bench_rccl1.err | grep '^{' > $R/bench_bf16x2_rccl_one_rank_one_graph.json
DPN_BENCH_ONE_DEVICE=1 DPN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-power 2> $R/bench_2ranks.err | grep '^{' > $R/bench_2ranks_one_device_gloo.json
DPN_BENCH_ONE_DEVICE=1 DPN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --leads 3 --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-power 2>> $R/bench_2ranks.err | grep '^{' > $R/bench_2ranks_one_device_gloo_3leads.json
timeout 600 python tools/tiles_timeline.py 37265 tl > $R/fwd_tiles_kernel_timeline.txt 2>&1
for v in tl tlnostore tlnomfma tlnoaload tlnosincos; do
  timeout 300 python tools/bwd_tiles_timeline.py 37265 $v > $R/bwd_tiles_timeline_$v.txt 2>&1
done
# the k-step in hand-scheduled assembly: variants, then clock + socket power of a few of them held for 4 s each
timeout 300 ./tools/microbench/kstep_asm > $R/microbench_kstep_asm.txt 2>&1
( for v in "MFMAs only, fragments in VGPRs" "LDS reads only, fragments in VGPRs" "L2 loads only, fragments in VGPRs" "VGPRs, accumulators in VGPRs, interleaved, spread" "VGPRs, accumulators in VGPRs, burst, chain" "AGPRs, accumulators in VGPRs, interleaved, spread" "1 tile(s) x 4" "4 tile(s) x 4"; do
  timeout 120 python tools/clock_watch.py "$v" -- ./tools/microbench/kstep_asm "$v" 4
done ) 2>&1 | grep -v "fclk\|mclk\|socclk\|level" > $R/microbench_kstep_asm_clocks.txt
# clock + socket power of the point kernels run back to back (5 s each), of the no-sincos ablation builds, and of the step
( for k in fwd bwd wgrad; do
  DPN_PROBE_SOAK=$k,5 timeout 200 python tools/clock_watch.py "point kernel $k, bf16x2, back to back" -- python tools/kernel_probe.py bf16x2
done
DPN_PROBE_SOAK=fwd,5 timeout 200 python tools/clock_watch.py "point kernel fwd, plain bf16, back to back" -- python tools/kernel_probe.py bf16
for v in tl tlnosincos tlnomfma tlnoaload tlnostore; do for k in fwd bwd; do
  DPN_LIB=$PWD/deepphysinet_amd/libdpn_hip_$v.so DPN_PROBE_SOAK=$k,4 timeout 200 python tools/clock_watch.py "point kernel $k, bf16x2, experiment build $v, back to back" -- python tools/kernel_probe.py bf16x2
done; done
timeout 300 python tools/clock_watch.py "bench.py, 3000 steps" -- python bench.py --steps 3000 --no-cpu-baseline --no-alt --no-power
) 2>&1 | grep -v "fclk\|mclk\|socclk\|level\|amdgpu.ids" | cut -c1-260 > $R/point_kernel_clocks.txt
timeout 300 python tools/enc_timeline.py > $R/enc_timeline_with_l2_helpers.txt 2>&1
DPN_ENC_NO_HELPERS=1 timeout 300 python tools/enc_timeline.py > $R/enc_timeline_cold_l2.txt 2>&1
DPN_ENC_NO_HELPERS=1 ENC_TL_WARM=1 timeout 300 python tools/enc_timeline.py > $R/enc_timeline_warm_l2.txt 2>&1
timeout 300 python tools/wgrad16_bench.py > $R/wgrad16_bench.txt 2>&1
( timeout 1200 python tools/soak.py bf16x2 300; timeout 900 python tools/soak.py bf16 300 ) 2>&1 | grep -v "amdgpu.ids\|AccumulateGrad\|run_backward" > $R/soak_bitwise.txt
timeout 600 python tools/enc_batch_check.py 1 3 > $R/encoder_vs_fp64.txt 2>&1
timeout 600 python tools/enc_batch_check.py 8 4 >> $R/encoder_vs_fp64.txt 2>&1
timeout 900 python tools/enc_batch_check.py 61 4 >> $R/encoder_vs_fp64.txt 2>&1
timeout 600 python tools/cfg2_debug.py 61 4 > $R/cfg2_numerics_fused.txt 2>&1
DPN_ENCODER_UNFUSED=1 timeout 600 python tools/cfg2_debug.py 61 4 > $R/cfg2_numerics_per_gemm_encoder.txt 2>&1
rm -rf $R/prof_cfg2; timeout 900 rocprofv3 --kernel-trace --stats -d $R/prof_cfg2 -o trace -- python3 bench.py --leads 61 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-power > $R/bench_prof_cfg2.log 2>&1
timeout 600 python tools/prof_summary.py $(find $R/prof_cfg2 -name "*.db" | head -1) 40 > $R/cfg2_61leads_kernel_stats_bf16x2.txt
for prec in bf16x2 bf16; do
  timeout 600 python tools/phase_times.py $prec > $R/phase_times_$prec.txt 2>&1
  DPN_ENC_NO_HELPERS=1 timeout 600 python tools/phase_times.py $prec > $R/phase_times_${prec}_no_l2_helpers.txt 2>&1
  DPN_ENCODER_UNFUSED=1 timeout 600 python tools/phase_times.py $prec > $R/phase_times_${prec}_per_gemm_encoder_of_round3.txt 2>&1
  timeout 600 python tools/reference_step.py $prec > $R/reference_shaped_step_$prec.json 2>> $R/refstep.err
  rm -rf $R/prof_$prec; timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof_$prec -o trace -- python3 bench.py --steps 10 --warmup 3 --prec $prec --no-cpu-baseline --no-alt --no-power > $R/bench_prof_$prec.log 2>&1
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
