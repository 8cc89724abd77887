"""Analyse a rocprofv3 kernel trace of bench.py: per-kernel time per step under lane concurrency."""
import csv, glob, collections, sys
f = (glob.glob(sys.argv[1] + '/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[0]
first, nsteps = int(sys.argv[2]), int(sys.argv[3])   # index of first timed conv0 launch, number of steps
rows = list(csv.DictReader(open(f)))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('crnn::', ''),
       r['Queue_Id'], int(r['Grid_Size_X'])) for r in rows]
ev.sort()
c0 = [e for e in ev if e[2] == 'conv0_pool_kernel']
t0, t1 = c0[first][0], c0[first + nsteps][0]
sel = [e for e in ev if t0 <= e[0] < t1]
wall = (t1 - t0) / 1e6
print('wall for %d steps %.2f ms => %.3f ms/step' % (nsteps, wall, wall / nsteps))
dur = collections.defaultdict(float); cnt = collections.Counter()
for e in sel:
    key = e[2] + (':%d' % e[4] if e[2] == 'conv_gemm_kernel' else '')
    dur[key] += (e[1] - e[0]) / 1e6; cnt[key] += 1
tot = 0
for k, v in sorted(dur.items(), key=lambda x: -x[1]):
    print('%-40s n=%3d  %.3f ms/step  avg %.3f' % (k, cnt[k], v / nsteps, v / cnt[k])); tot += v
print('sum of durations per step %.2f ms' % (tot / nsteps))
iv = sorted((e[0], e[1]) for e in sel)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print('union busy %.2f ms of %.2f' % (busy / 1e6, wall))
print('queues', collections.Counter(e[3] for e in sel))
