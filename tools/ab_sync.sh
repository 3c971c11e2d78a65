for i in 1 2 3; do for CFG in "0 1" "1 4" "0 2"; do set -- $CFG
  RTGPU_SYNC_LEVELS=$1 RTGPU_SIDE_STREAMS=$2 timeout -k 10 120 python bench.py --steps 32 --warmup 8 --cpu-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sync_levels=$1 side=$2', round(d['ms_per_step'],3), round(d['ms_per_frame_alone'],3))" || exit 1
done; done
