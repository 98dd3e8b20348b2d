cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_model.py -x -q -k train_step 2>&1 | grep -B5 "Error" | head -40
