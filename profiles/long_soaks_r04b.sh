# second, longer parity run on the final round-4 tree (other seeds)
python3 profiles/prefilter_soak.py 1200 411 > gpurun_out/r04_long_soak_prefilter_b.txt 2>/dev/null
python3 tests/fuzz_gpu.py 900 412 > gpurun_out/r04_long_fuzz_b.txt 2>/dev/null
tail -1 gpurun_out/r04_long_soak_prefilter_b.txt; tail -1 gpurun_out/r04_long_fuzz_b.txt
