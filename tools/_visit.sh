set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "test_gpu_model or test_gpu_parallel or bf16_step or step0 or drivers" 2>&1 | tail -5
export AB_CONFIGS="SSD_WGRAD_ROWS_WGS_BF16=2048;SSD_WGRAD_ROWS_WGS_BF16=1024;SSD_WGRAD_ROWS_WGS_BF16=1536"
bash tools/ab_step.sh r02_n bf16
AB_CONFIGS="SSD_NOP=1" bash tools/ab_step.sh r02_n f32
