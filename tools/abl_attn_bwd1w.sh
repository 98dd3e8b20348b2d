# usage (GPU box): bash tools/abl_attn_bwd1w.sh  -- the in-tree library and every build_ab/liboctmae_abl_*.so (make -C octcubem_amd/csrc abl)
cd $GRAFT_REPO_ROOT
python tools/attn_bwd1w_time.py
for so in build_ab/liboctmae_abl_*.so; do OCTMAE_LIB=$GRAFT_REPO_ROOT/$so python tools/attn_bwd1w_time.py; done
python tools/attn_bwd1w_time.py
