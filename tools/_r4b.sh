set -o pipefail
mkdir -p gpurun_out/r4b
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4b/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4b/summary.txt
export KF_LIB=$PWD/hybkinectfu_amd/libhybkf_exp.so KF_INTEGRATE_EXP=13
for cfg in c2 c4; do timeout -k 10 300 python tools/exp_wave_kinds.py $cfg 200 > gpurun_out/r4b/kinds_$cfg.txt 2>&1; echo "kinds $cfg rc=$?" | tee -a gpurun_out/r4b/summary.txt; done
timeout -k 10 300 python tools/exp_wave_kinds.py c5 24 > gpurun_out/r4b/kinds_c5.txt 2>&1; echo "kinds c5 rc=$?" | tee -a gpurun_out/r4b/summary.txt
