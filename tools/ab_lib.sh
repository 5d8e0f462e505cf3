# alternate two libraries on one box: bash tools/ab_lib.sh variants/libuic_x.so [rounds]
V=$1; R=${2:-3}
for i in $(seq $R); do
  python bench.py --steps 300 --warmup 30 --long-run 0 --no-f32 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base   ', d['ms_per_step'], d['per_image_features']['ms_per_step'])"
  UIC_LIB=$V python bench.py --steps 300 --warmup 30 --long-run 0 --no-f32 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant', d['ms_per_step'], d['per_image_features']['ms_per_step'])"
done
