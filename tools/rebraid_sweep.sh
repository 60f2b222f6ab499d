# tools/rebraid_sweep.sh: the two-level forest (C4 as instanced geometry) over re-braiding widths and collapse rules (diffuse BSDF, as profiles/*_c4_two_level)
run() {
  env "$@" timeout 600 python bench.py --scene forest --flatten 0 --steps 40 --warmup 4 --no-cpu-baseline --sustained-seconds 0 --static-camera 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['roofline']['counts_per_step']
        print('%-60s ms/step %.3f  nodes/closest ray %.2f  shadow %.2f  tris %.2f' % ('$*', d['ms_per_step'], c['nodes_closest']/c['rays_closest'], c['nodes_shadow']/max(c['rays_shadow'],1), c['tris_closest']/c['rays_closest']))
"
}
run RPTR_REBRAID=4
run RPTR_REBRAID=1
run RPTR_REBRAID=2
run RPTR_REBRAID=4 RPTR_TLAS_COLLAPSE=optimal
run RPTR_REBRAID=16 RPTR_TLAS_COLLAPSE=optimal
run RPTR_REBRAID=16 RPTR_COLLAPSE=optimal
run RPTR_REBRAID=64 RPTR_COLLAPSE=optimal RPTR_TLAS_COLLAPSE=optimal
run RPTR_REBRAID=256 RPTR_COLLAPSE=optimal RPTR_TLAS_COLLAPSE=optimal
