cd $GRAFT_REPO_ROOT
python profiles/fuzz_case.py profiles/cases/fuzz_fail_2570_1607.npz 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_fuzz_case.txt; cat gpurun_out/r06_fuzz_case.txt
timeout 300 python tests/fuzz_gpu.py 60 5001 > gpurun_out/r06_fuzz_debug.txt 2>&1; tail -c 400 gpurun_out/r06_fuzz_debug.txt
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_gpu_suite.txt; cat gpurun_out/r06_gpu_suite.txt
