# developer A/B runs of bench.py under different recall policies (environment switches of recall.hip)
run() { # label, batch, env...
  label=$1; batch=$2; shift 2
  env "$@" python bench.py --steps 10 --warmup 2 --batch $batch --no-cpu-baseline --latency-reqs 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('$label', 'batch', $batch, 'value %.4g' % j['value'], 'ms/step %.2f' % j['ms_per_step'], 'scan ms %.2f' % r['ms_per_pass'], 'frac %.3f' % r['frac'])"
}
for spec in "$@"; do
  label=${spec%%:*}; rest=${spec#*:}; batch=${rest%%:*}; envs=${rest#*:}
  run $label $batch $envs
done
