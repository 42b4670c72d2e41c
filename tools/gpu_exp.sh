cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for mb in 0 1 2; do
  FIL_CIN_MB=$mb timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('MB $mb ms/step %.3f fwd_l1 %.4f dz_l1 %.4f dw_l1 %.4f fwd_l2 %.4f dz_l2 %.4f last %.4f'%(d['ms_per_step'], k['cin_fwd_l1']['avg_ms'], k['cin_bwd_dz_l1']['avg_ms'], k['cin_bwd_dw_l1']['avg_ms'], k['cin_fwd_l2']['avg_ms'], k['cin_bwd_dz_l2']['avg_ms'], k['cin_last_bwd']['avg_ms']))"
done
done
