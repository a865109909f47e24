for s in 0 8 64; do
  python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --consensus-seconds 0 --profile-stride $s 2>&1 | tail -1 > /tmp/b_$s.json
  python -c "
import json; d=json.load(open('/tmp/b_$s.json')); print('stride', $s, round(d['value']), round(d['ms_per_step'],2), round(d['loop_ms_events_per_step'],2), d['roofline']['avg_launch_ms'], d['roofline']['launches_timed'])"
done
