# copy the summaries of tools/refresh_profiles.sh (gpurun_out/r4) into profiles/ (tracked), named per round
set -e
S=gpurun_out/r4; D=profiles
for f in bench_bf16x2.json bench_bf16.json bench_bf16x2_per_gemm_encoder_of_round3.json bench_cfg2_61leads_bf16x2.json bench_cfg2_61leads_bf16.json \
         bench_cfg2_61leads_bf16x2_per_gemm_encoder_NaN_state.json bench_bf16x2_rccl_one_rank.json bench_bf16x2_rccl_one_rank_one_graph.json \
         bench_2ranks_one_device_gloo.json bench_2ranks_one_device_gloo_3leads.json fwd_tiles_kernel_timeline.txt enc_timeline_with_l2_helpers.txt \
         enc_timeline_cold_l2.txt enc_timeline_warm_l2.txt wgrad16_bench.txt encoder_vs_fp64.txt cfg2_numerics_fused.txt cfg2_numerics_per_gemm_encoder.txt \
         cfg2_61leads_kernel_stats_bf16x2.txt phase_times_bf16x2.txt phase_times_bf16.txt phase_times_bf16x2_no_l2_helpers.txt \
         phase_times_bf16x2_per_gemm_encoder_of_round3.txt reference_shaped_step_bf16x2.json reference_shaped_step_bf16.json \
         kernel_trace_stats_bench_bf16x2.txt kernel_trace_stats_bench_bf16.txt step_timeline_bf16x2.txt step_timeline_bf16.txt \
         pmc_eager_step_bf16x2.txt pmc_eager_step_bf16.txt microbench_kstep_asm.txt microbench_kstep_asm_clocks.txt point_kernel_clocks.txt soak_bitwise.txt; do
  [ -f $S/$f ] && grep -v "amdgpu.ids\|AccumulateGrad\|run_backward" $S/$f > $D/round4_$f
done
for v in tl tlnostore tlnomfma tlnoaload tlnosincos; do
  n=$(echo $v | sed 's/^tl$/full/; s/^tl//'); grep -v "amdgpu.ids" $S/bwd_tiles_timeline_$v.txt > $D/round4_bwd_tiles_timeline_$n.txt
done
cp $S/pmc_traffic.json $D/pmc_traffic.json
ls $D | grep round4 | wc -l
