#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: per step, when the alpha CG (the queue that runs k_pcg_update_xr) starts and
ends relative to the posterior's float64 GEMMs -- how much of the CG is exposed after the covariance work."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def nm(k):
    m = re.search(r'(k_[a-z0-9_]+)', k); return m.group(1) if m else k[:30]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['n'] = nm(r['Kernel_Name'])
cgq = collections.Counter(r['Queue_Id'] for r in rows if r['n'] == 'k_pcg_update_xr').most_common(1)[0][0]
cg = sorted([r for r in rows if r['Queue_Id'] == cgq], key=lambda r: r['s'])
segs, cur = [], [cg[0]]
for r in cg[1:]:
    if r['s'] - cur[-1]['e'] > 15e6: segs.append(cur); cur = [r]
    else: cur.append(r)
segs.append(cur)
other = sorted([r for r in rows if r['Queue_Id'] != cgq], key=lambda r: r['s'])
for sg in segs:
    s0, e0 = sg[0]['s'], sg[-1]['e']
    f64 = [r for r in other if r['n'] == 'k_gemm_nt_f64' and s0 - 30e6 < r['s'] < e0 + 120e6]
    potrf_end = max([r['e'] for r in other if r['n'] in ('k_gemm_nt_h3', 'k_potrf_leaf', 'k_triinv_leaf') and r['e'] <= s0 + 2e6] or [s0])
    last = max([r['e'] for r in other if s0 < r['e'] < e0 + 120e6 and (not f64 or r['e'] <= max(x['e'] for x in f64) + 15e6)] or [e0])
    print("CG: %d kernels, span %.1f ms (starts %.1f ms after the last factor kernel); float64 GEMMs run %.1f .. %.1f ms; "
          "last other kernel of the step ends at %.1f ms; CG ends at %.1f ms" % (
              len(sg), (e0 - s0) / 1e6, (s0 - potrf_end) / 1e6, (min(r['s'] for r in f64) - s0) / 1e6 if f64 else -1,
              (max(r['e'] for r in f64) - s0) / 1e6 if f64 else -1, (last - s0) / 1e6, (e0 - s0) / 1e6))
