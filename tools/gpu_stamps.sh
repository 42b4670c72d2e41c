cd $GRAFT_REPO_ROOT
rm -f ml_function_amd/build/attn*.o
FIL_HIPCC_FLAGS=-DFIL_ATTN_STAMPS python -m ml_function_amd.build > /dev/null 2>&1
python tools/attn_stamps.py 1 2>&1 | tail -12
# back to the normal build (the flags stamp of ml_function_amd/build.py forces the rebuild)
python -m ml_function_amd.build > /dev/null 2>&1
