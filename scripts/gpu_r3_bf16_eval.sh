cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 500 -p no:cacheprovider -k "eval_800" > gpurun_out/pytest_gpu_eval.log 2>&1; tail -12 gpurun_out/pytest_gpu_eval.log
timeout 200 python bench.py --mode eval --no-cpu-baseline > gpurun_out/bench_eval_fp32.log 2>&1; grep "^{" gpurun_out/bench_eval_fp32.log | cut -c1-400
timeout 200 python bench.py --mode eval --no-cpu-baseline --precision bf16x3 > gpurun_out/bench_eval_bf16x3.log 2>&1; grep "^{" gpurun_out/bench_eval_bf16x3.log | cut -c1-600; tail -3 gpurun_out/bench_eval_bf16x3.log | cut -c1-300
timeout 200 python bench.py --mode eval --no-cpu-baseline --precision bf16x3 --hidden 128 --layers 4 > gpurun_out/bench_eval_bf16x3_4x128.log 2>&1; grep "^{" gpurun_out/bench_eval_bf16x3_4x128.log | cut -c1-300
timeout 200 python bench.py --mode eval --no-cpu-baseline --hidden 128 --layers 4 > gpurun_out/bench_eval_fp32_4x128.log 2>&1; grep "^{" gpurun_out/bench_eval_fp32_4x128.log | cut -c1-300
