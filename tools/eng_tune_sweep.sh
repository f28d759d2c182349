for t in 0x880 0x881 0x480 0x080 0x840 0x8c0 0x841 0xc80; do
  echo -n "CM_ENG_TUNE=$t: "; CM_ENG_TUNE=$t timeout 120 python bench.py --no-cpu-baseline --steps 96 --warmup 8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline_step']['frac'])"
done
