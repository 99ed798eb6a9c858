timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "last_layer_on_head or development_library" > gpurun_out/r06e_tests.log 2>&1; tail -5 gpurun_out/r06e_tests.log
bash tools/ab_train.sh r06e 33=0 33=1
