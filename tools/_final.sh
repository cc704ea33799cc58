cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/final_pytest.txt 2>&1
for i in 1 2 3; do python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_model.py -q -k "pipeline or se_ or xcd or graph or sub_batch" >> gpurun_out/final_repeat.txt 2>&1; done
bash tools/round_profile.sh r02 > /dev/null 2>&1
bash tools/valu.sh r02 valu --batch 64 > /dev/null 2>&1
for f in kernel_stats.csv kernel_stats_one_forward.csv hbm_traffic.json mfma_util.json per_op.txt; do cp gpurun_out/r02/$f profiles/r02_$f; done
bash tools/_lines.sh
rm -f gpurun_out/r02b/batch_sweep.txt; mkdir -p gpurun_out/r02b
for bs in 1 8 16 32 64 128 256; do for fl in 3 1; do python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 200 --warmup 20 --batch $bs --inflight $fl 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $bs  forwards in flight $fl  %.1f img/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" >> gpurun_out/r02b/batch_sweep.txt; done; done
