"""GPU busy time per step from a rocprofv3 kernel trace (csv): steps are delimited by the optimizer kernel (adamw_k); over the last N
steps: wall time per step, summed kernel time per step, launches per step, and the idle share.  Usage: busy.py <kernel_trace.csv> [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
name = 'Kernel_Name' if 'Kernel_Name' in rows[0] else 'Name'
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r[name]) for r in rows))
marks = [e for s, e, n in ks if 'adamw_k' in n]
assert len(marks) > N, len(marks)
t0, t1 = marks[-N - 1], marks[-1]
win = [(s, e, n) for s, e, n in ks if s >= t0 and e <= t1]
busy = sum(e - s for s, e, n in win)
# union of intervals (kernels may overlap)
union, cur_s, cur_e = 0, None, None
for s, e, n in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += (cur_e - cur_s) if cur_e else 0
print(f'last {N} steps: {(t1 - t0) / N / 1e6:.3f} ms/step wall, {busy / N / 1e6:.3f} ms/step summed kernel time, {union / N / 1e6:.3f} ms/step GPU non-idle, '
      f'{len(win) / N:.0f} launches/step, idle {(1 - union / (t1 - t0)) * 100:.1f} %')
short = sum(1 for s, e, n in win if e - s < 10000)
print(f'   kernels shorter than 10 us: {short / N:.0f} per step; mean gap between consecutive kernels {((t1 - t0) - union) / max(1, len(win)) / 1e3:.2f} us')

import collections, re
agg = collections.defaultdict(lambda: [0, 0])
for s_, e_, n_ in win:
    key = re.sub(r'<.*', '', n_.replace('void ', ''))[:70]
    agg[key][0] += e_ - s_; agg[key][1] += 1
print('   top kernels (ms/step, launches/step):')
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f'      {t / N / 1e6:7.3f} {c / N:7.1f}  {k}')
tiny = sum(e_ - s_ for s_, e_, n_ in win if e_ - s_ < 10000)
print(f'   time inside kernels shorter than 10 us: {tiny / N / 1e6:.3f} ms/step')
