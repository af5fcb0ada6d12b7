for rep in 1 2; do for n in 128 256; do for e in 4194304 0; do
python tools/mech_bench.py $n 2 0 $e 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('n',j['n'],'exp',j['exp'],'ms/it %.4f'%j['ms_per_cg_iteration'],'cg',j['cg_its'],' '.join('%s %.4f'%(k['kernel'].replace('gamma_',''),k['avg_ms']) for k in j['kernels'][:7]))"
done; done; done
