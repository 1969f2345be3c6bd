for args in "--scheme 2s" "--scheme 2s --dtype f32" "--scheme n79" "--scheme n79 --dtype f32" "--scheme zq" "--scheme zq --dtype f32" "--scheme zq_pa" "--scheme zq --nz 100 --ncol 6000" "--scheme zq --nb 38 --nz 100 --ncol 100000 --steps 10 --warmup 3" "--scheme n79 --nb 107 --ncol 30000" "--scheme 2s --nb 107 --ncol 30000" "--scheme zq --nb 107 --ncol 30000" "--scheme zq_pa --nb 107 --ncol 30000" "--scheme zq_pa --nb 38 --nz 100 --ncol 100000 --steps 10 --warmup 3" "--scheme 2s --nb 38 --ncol 200000 --steps 10 --warmup 3" "--scheme 4s" "--scheme bf" "--scheme g77" "--scheme bl"; do
  python bench.py $args --repeats 3 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('$args'.ljust(70), 'ms/step', round(d['ms_per_step'], 4), 'kernel', round(r['kernel_ms_avg'], 4), 'frac', round(r['frac'], 3), r['kernel'][:50])
"
done
