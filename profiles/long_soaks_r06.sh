# long parity runs on the round-6 tree (one gpurun call): the per-tile band rule with the packed scan against the plain kernel, the fuzz of all paths through the
# PRODUCT library, the exact matcher's polled merge against the pre-filter matcher
python3 profiles/prefilter_soak.py ${1:-900} 501 > gpurun_out/r06_long_soak_prefilter.txt 2>/dev/null
python3 tests/fuzz_gpu.py ${2:-420} 502 > gpurun_out/r06_long_fuzz.txt 2>/dev/null
python3 profiles/match_soak.py ${3:-200} 504 > gpurun_out/r06_long_soak_match.txt 2>/dev/null
tail -1 gpurun_out/r06_long_soak_prefilter.txt; tail -1 gpurun_out/r06_long_fuzz.txt; tail -1 gpurun_out/r06_long_soak_match.txt
