# builds libmmfusion with other chunk / workgroup sizes of the training attention kernels and times them (GPU box)
cd "$GRAFT_REPO_ROOT/nvblox_mindmap_amd/csrc"
for cfg in "128 8" "256 8" "128 4" "64 8" "128 16"; do
  set -- $cfg
  rm -f _build/mmf_kernels_train_attn.o
  make -s CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DMMF_TA_CHUNK=$1 -DMMF_TA_WAVES=$2" > /dev/null 2>&1
  echo "chunk $1 waves $2: $(cd ../.. && python3 tools/microbench/train_attention_check.py 2>&1 | grep 'mine fwd')"
done
rm -f _build/mmf_kernels_train_attn.o; make -s > /dev/null 2>&1
