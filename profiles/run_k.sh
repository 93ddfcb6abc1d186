for k in 1 2 3 4 5 6 8; do
  python bench.py --light --steps 30 --preset fast --k $k --refs 56 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('k %d  kernel_ms %.4f  ps per cell %.2f  surv %.4f  frac %.3f' % ($k, r['kernel_ms'], r['kernel_ms'] * 1e9 / (56 * 512 * 512), d['survivor_fraction'], r['frac']))
"
done
