for i in 1 2; do
for v in 22 30; do
SLIC_CONV_BIG_VARIANT=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
