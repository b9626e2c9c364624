# copy the summaries of tools/refresh_profiles.sh (gpurun_out/r6) into profiles/ (tracked), named per round.  Nothing is filtered out of the files but the
# box's missing-file notice (amdgpu.ids); a file that is absent is reported, not silently skipped.
S=gpurun_out/r6; D=profiles
for f in bench_bf16x2.json bench_bf16x2_driver_protocol.json bench_bf16.json bench_cfg2_61leads_bf16x2.json bench_cfg2_61leads_bf16.json bench_cfg4_encoder_fp8_mx.json \
         bench_bf16x2_rccl_one_rank.json bench_bf16x2_rccl_one_rank_one_graph.json bench_bf16x2_rccl_one_rank_form_trial.json bench_bf16x2_rccl_one_rank_trial_stall_fallback.json \
         bench_2ranks_one_device_gloo.json bench_2ranks_one_device_gloo_3leads.json bench_8ranks_one_device_gloo_4096pts.json \
         fwd_tiles_kernel_timeline.txt bwd_tiles_timeline_full.txt fwd_pp_vs_tiles.txt fwd_pp_kernel_timeline.txt encoder_vs_fp64.txt cfg2_61leads_kernel_stats_bf16x2.txt phase_times_bf16x2.txt phase_times_bf16.txt \
         reference_shaped_step_bf16x2.json reference_shaped_step_bf16.json kernel_trace_stats_bench_bf16x2.txt kernel_trace_stats_bench_bf16.txt \
         step_timeline_bf16x2.txt step_timeline_bf16.txt pmc_eager_step_bf16x2.txt pmc_eager_step_bf16.txt soak_bitwise.txt tests_gpu.txt; do
  if [ -f $S/$f ]; then grep -v "amdgpu.ids" $S/$f > $D/round6_$f || true; else echo "MISSING: $S/$f"; fi
done
[ -f $S/pmc_traffic.json ] && cp $S/pmc_traffic.json $D/pmc_traffic.json || echo "MISSING: $S/pmc_traffic.json"
ls $D | grep round6 | wc -l
