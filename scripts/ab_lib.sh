#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: scripts/ab_lib.sh out.log path/to/other.so [bench args...]   (W3D_LIB_PATH selects the library: _lib.py)
out=$1; other=$2; shift 2
: > "$out"
run() {
  W3D_LIB_PATH=$2 python bench.py --steps 20 --warmup 5 --no-side-lines --no-cpu-baseline --whole-run 0 "${@:3}" 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read())
st = {r['stage'].split(' (')[0][:28]: (round(r.get('ms_in_loop') or 0, 3), round(r['ms'], 3)) for r in o['rooflines'] if 'ms_in_loop' in r}
print(json.dumps({'lib': sys.argv[1], 'ms_per_step': round(o['ms_per_step'], 3), 'knn_ms': round(o['roofline']['step_ms'], 3), 'swept': round(o['roofline']['swept_fraction'], 4), 'frac': round(o['roofline']['frac'], 4), 'stages[in_loop,alone]': st}))" "$1" >> "$out"
}
new=$(pwd)/wast3d_amd/libwast3d_hip.so
for i in 1 2; do run other "$(pwd)/$other" "$@"; run new "$new" "$@"; done
cat "$out"
