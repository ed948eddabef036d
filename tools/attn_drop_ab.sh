# A/B of the dropout variants' occupancy settings (rebuilds attention.o per setting on the GPU box)
set -e
cd $GRAFT_REPO_ROOT/tethys-speech_amd/csrc
python3 ../../tools/attn_bench.py | head -4
for cfg in "3 2" "3 3" "2 2"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -DTMI_ATTN_DROP_OCC=$1 -DTMI_ATTN_DQ_DROP_OCC=$2 -c attention.hip -o attention.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtethys_mi.so runtime.o gemm.o gemm_fast.o layernorm.o softmax_xent.o adam_misc.o attention.o wav2vec2.o
  echo "== fwd occ $1, dq occ $2"
  ATTN_DROPOUT=0.1 python3 ../../tools/attn_bench.py | head -4
done
