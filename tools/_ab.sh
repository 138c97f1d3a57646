run() { python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-torch-adam --no-probe 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
RU3D_WGRAD_PAIR=1 run pair
RU3D_WGRAD_PAIR=0 run nopair
done
