cd $GRAFT_REPO_ROOT
for sp in 0 30 33 36 39 42 45 48 52 58 64; do
  FIL_CIN_DW_SPLITS=$sp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('splits $sp ms/step %.3f dw_l1 %.4f dw_l2 %.4f reduce %.4f last %.4f'%(d['ms_per_step'], k['cin_bwd_dw_l1']['avg_ms'], k['cin_bwd_dw_l2']['avg_ms'], k['cin_reduce_dw']['avg_ms'], k['cin_last_bwd']['avg_ms']))"
done
