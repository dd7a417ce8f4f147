"""One-line view of a bench.py JSON line: `python tools/show_bench.py FILE...`."""
import json, sys
for p in sys.argv[1:]:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    r = d.get('roofline', {})
    print('%s: %.0f %s, %.2f ms/step, kernel %s %.3f ms, frac %.4f, redone %s' % (
        p, d['value'], d['unit'], d['ms_per_step'], r.get('kernel'), r.get('kernel_ms_per_launch') or 0,
        r.get('frac') or 0, d['config'].get('reads_redone_exact')))
