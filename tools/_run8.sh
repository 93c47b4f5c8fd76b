set -e
bash tools/profile_r04.sh stats sq traffic headline n2 u238 library scaling
python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_final.log 2>&1 || { tail -40 gpurun_out/r04/pytest_gpu_final.log; exit 1; }
tail -3 gpurun_out/r04/pytest_gpu_final.log
