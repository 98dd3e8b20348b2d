cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "hd32_both_forms or long_sequences" 2>&1 | tail -2
for i in 1 2; do
python tools/attn_bwd1w_time.py
OCTMAE_LIB=$GRAFT_REPO_ROOT/build_ab/liboctmae_pkmul.so python tools/attn_bwd1w_time.py
done 2>&1 | grep median | tee gpurun_out/r3g/pkmul.txt
