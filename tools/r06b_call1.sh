tools/probe/valu_rate.bin > gpurun_out/r06b_valu_rate.txt 2>&1
bash tools/flash_prio_ab.sh > gpurun_out/r06b_flash_prio_ab.txt 2>&1
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "sum_of_squares or gemm_epilogues or gemm_full_tile or kmajor_operands" > gpurun_out/r06b_t1.log 2>&1
python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "fused_gradient_norm or force_dist or train_step" > gpurun_out/r06b_t2.log 2>&1
for rep in 1 2; do
  OTTER_NO_FUSED_GRAD_NORM=1 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06b_norm_sweep_$rep.json
  python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r06b_norm_fused_$rep.json
done
tail -3 gpurun_out/r06b_t1.log gpurun_out/r06b_t2.log
