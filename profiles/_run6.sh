cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; tail -1 gpurun_out/r06_smoke.txt
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 > gpurun_out/r06_gpu_suite.txt; cat gpurun_out/r06_gpu_suite.txt
python bench.py > gpurun_out/r06_bench_final_check.json 2> /dev/null; head -c 600 gpurun_out/r06_bench_final_check.json
