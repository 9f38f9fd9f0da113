python -m pytest tests/test_gpu_pose.py tests/test_gpu_dino.py tests/test_gpu_facade.py tests/test_pairs_stream.py tests/test_gpu_ransac.py -x -q -m gpu > gpurun_out/t.txt 2>&1; grep -a "passed\|failed\|rror" gpurun_out/t.txt | tail -3
for i in 1 2; do
  echo "== new"; sh profiles/c1_timeline.sh 1024
  echo "== old"; SFM_AMD_LIB_DIR=$PWD/build/old_lib sh profiles/c1_timeline.sh 1024
done
for i in 1 2; do python profiles/c1_ab.py 2>/dev/null | tail -1; SFM_AMD_LIB_DIR=$PWD/build/old_lib python profiles/c1_ab.py 2>/dev/null | tail -1; done
