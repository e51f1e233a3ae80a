cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_vgg_gpu.py tests/test_train_gpu.py tests/test_backward_gpu.py -x -q > gpurun_out/train_pytest.log 2>&1; tail -5 gpurun_out/train_pytest.log
timeout 300 python tools/bench_train.py 2>&1 | grep step
bash tools/_tr.sh
