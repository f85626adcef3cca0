for ex in ${EXPS:-0 4}; do
  GPSX_MX_EXPERIMENT=$ex python bench.py --no-cpu-baseline --no-tracking --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ex', $ex, 'ms', round(d['roofline']['kernel_ms'],3))
" 2>&1 | tail -1
done
