# long parity runs on the final round-4 tree (one gpurun call)
python3 profiles/prefilter_soak.py 420 401 > gpurun_out/r04_long_soak_prefilter.txt 2>/dev/null
python3 tests/fuzz_gpu.py 420 402 > gpurun_out/r04_long_fuzz.txt 2>/dev/null
SECONDS=200 SEED=403 python3 profiles/match_fused_soak.py > gpurun_out/r04_long_soak_match_fused.txt 2>/dev/null
python3 profiles/match_soak.py 200 404 > gpurun_out/r04_long_soak_match.txt 2>/dev/null
tail -1 gpurun_out/r04_long_soak_prefilter.txt; tail -1 gpurun_out/r04_long_fuzz.txt; tail -1 gpurun_out/r04_long_soak_match_fused.txt; tail -1 gpurun_out/r04_long_soak_match.txt
