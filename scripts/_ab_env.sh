# usage: _ab_env.sh out.log VAR v1 v2 v3 ...
out=$1; var=$2; shift 2
: > $out
for rep in 1 2; do for v in "$@"; do
  env $var=$v python bench.py --steps 20 --warmup 5 --no-side-lines --no-cpu-baseline --whole-run 0 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read())
st = {r['stage'].split(' (')[0][:20]: round(r.get('ms_in_loop') or 0, 3) for r in o['rooflines'] if 'ms_in_loop' in r}
print(json.dumps({'v': sys.argv[1], 'ms_per_step': round(o['ms_per_step'], 3), 'knn_ms': round(o['roofline']['step_ms'], 3), 'in_loop': st}))" $v >> $out
done; done
cat $out
