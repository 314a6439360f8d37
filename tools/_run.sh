set -e
python -m pytest tests -m gpu -x -q > gpurun_out/s2_tests.log 2>&1 || { tail -30 gpurun_out/s2_tests.log; exit 1; }
tail -3 gpurun_out/s2_tests.log
python tools/exp_epilogue.py > gpurun_out/s2_sched.log 2>&1; cat gpurun_out/s2_sched.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/s2_bench.log 2>&1; tail -1 gpurun_out/s2_bench.log | cut -c1-400
