for PR in 0 1; do for P in 8 4; do
  GGL_HUB_PRIORITY=$PR timeout 600 python bench.py --dry-parts $P --steps 10 --warmup 3 --no-comparison --no-cpu-baseline --pmc-traffic off --secondary off 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hub_priority=$PR dry P=$P ms_per_step', round(d['ms_per_step'],3))"
done; done
for PR in 0 1; do GGL_HUB_PRIORITY=$PR timeout 600 python bench.py --steps 10 --warmup 3 --no-comparison --no-cpu-baseline --pmc-traffic off --secondary off 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hub_priority=$PR products ms_per_step', round(d['ms_per_step'],3), 'agg', round(d['roofline']['ms_per_aggregate'],3))"; done
