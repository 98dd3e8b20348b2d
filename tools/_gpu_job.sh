cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
bash tools/abl_attn_bwd1w.sh > gpurun_out/r3d/abl.txt 2>&1; grep median gpurun_out/r3d/abl.txt
