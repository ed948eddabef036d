cd $GRAFT_REPO_ROOT
for abl in 0 1 2 3 4; do
  echo "== ABL=$abl"
  TMI_ATTN_V2=0 TMI_ATTN_ABL=$abl python tools/attn_bench.py 2>&1 | grep -E "enc-self   fwd"
done
