for cfg in "--k 3 --refs 64" "--k 4 --refs 64" "--k 5 --refs 56"; do for pad in 0 4352 69888 1052672; do
  LFD_BENCH_PLANE_PAD=$pad python bench.py --light --steps 30 --preset fast $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-18s pad %8d  kernel_ms %.4f  frac %.3f' % ('$cfg', $pad, r['kernel_ms'], r['frac']))
"
done; done
