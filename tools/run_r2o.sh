cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
timeout 1500 python3 -m pytest tests -m gpu -q -s -k "map_on_fixed or uint8_input_vs or image_mean_std or full_size" > gpurun_out/r2o/pytest.txt 2>&1; grep -E "mAP|passed|failed|FAILED|ground truth|classes whose|uint8 image|non-default|Error|error" gpurun_out/r2o/pytest.txt | head -30
