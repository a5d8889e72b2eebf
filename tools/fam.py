"""One line per bench log: frames/s, ms per step, time of the token-GEMM family and of the kernels named on the command line."""
import sys, json
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l); r = d['roofline']
        extra = " ".join("%s %.1f" % (k, r['also'][k]['total_us_per_step']) for k in sys.argv[2:] if k in r['also'])
        print(sys.argv[1], d['value'], d['ms_per_step'], 'family us', r['total_us_per_step'], extra)
