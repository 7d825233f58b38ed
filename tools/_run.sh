cd $GRAFT_REPO_ROOT
L=$PWD/opensearch-neural-pre-train_amd/snx
for lib in libsnx libsnx_pb libsnx_pc libsnx libsnx_pb; do
  echo "== $lib"
  SNX_LIB=$L/$lib.so CASES=store:2304:768,store:768:2304,rope_rows:2304:768,geglu_fwd:2304:768 ROUNDS=3 timeout -k 10 200 python3 tools/gpu_nt256.py time 2>&1 | grep " N="
done
