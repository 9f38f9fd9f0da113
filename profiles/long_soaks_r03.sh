python3 profiles/prefilter_soak.py 1200 201 > gpurun_out/r03_long_soak_prefilter.txt 2>/dev/null
python3 tests/fuzz_gpu.py 600 202 > gpurun_out/r03_long_fuzz.txt 2>/dev/null
SECONDS=600 SEED=203 python3 profiles/match_fused_soak.py > gpurun_out/r03_long_soak_match_fused.txt 2>/dev/null
python3 profiles/match_soak.py 300 204 > gpurun_out/r03_long_soak_match.txt 2>/dev/null
tail -1 gpurun_out/r03_long_soak_prefilter.txt; tail -1 gpurun_out/r03_long_fuzz.txt; tail -1 gpurun_out/r03_long_soak_match_fused.txt; tail -1 gpurun_out/r03_long_soak_match.txt
