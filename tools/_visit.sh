set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_s; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
for rep in 1 2; do for x in 0 1; do echo "SSD_C64_RING=$x"; SSD_C64_RING=$x timeout 100 python tools/bench_conv.py conv1_2 bf16 2>&1 | grep conv1_2; done; done
export AB_CONFIGS="SSD_C64_RING=0;SSD_C64_RING=1"
bash tools/ab_step.sh r02_s bf16
