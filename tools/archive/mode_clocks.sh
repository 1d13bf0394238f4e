# per-process modes of the step time against the GPU's clock / power read from sysfs while the bench runs (ten processes)
python - <<'P'
import glob, os
for r in glob.glob('/sys/class/drm/renderD*'):
    n = os.path.basename(r)
    if os.path.exists('/dev/dri/' + n):
        print('visible', n, os.path.realpath(r + '/device'))
        for f in sorted(glob.glob(r + '/device/hwmon/hwmon*/*_input') + glob.glob(r + '/device/hwmon/hwmon*/power1_average') + glob.glob(r + '/device/pp_dpm_sclk')):
            try: print('  ', f, open(f).read().strip().replace('\n', ' | ')[:200])
            except Exception as e: print('  ', f, 'unreadable', e)
P
cat > /tmp/sampler.py <<'P'
import glob, os, sys, time
dev = None
for r in glob.glob('/sys/class/drm/renderD*'):
    if os.path.exists('/dev/dri/' + os.path.basename(r)): dev = r + '/device'
files = glob.glob(dev + '/hwmon/hwmon*/freq1_input') + glob.glob(dev + '/hwmon/hwmon*/power1_average') + glob.glob(dev + '/hwmon/hwmon*/power1_input') + glob.glob(dev + '/hwmon/hwmon*/temp1_input')
out = open(sys.argv[1], 'w')
while True:
    vals = []
    for f in files:
        try: vals.append(open(f).read().strip())
        except Exception: vals.append('nan')
    out.write('%.3f %s\n' % (time.time(), ' '.join(vals))); out.flush()
    time.sleep(0.05)
P
for rep in 1 2 3 4 5 6 7 8 9 10; do
  python /tmp/sampler.py /tmp/samples_$rep.txt & SP=$!
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null > /tmp/bench_$rep.json
  kill $SP; wait $SP 2>/dev/null
  python - $rep <<'P'
import sys, json, numpy as np
rep = sys.argv[1]
d = json.loads(open('/tmp/bench_%s.json' % rep).read().strip().splitlines()[-1])
rows = [l.split() for l in open('/tmp/samples_%s.txt' % rep)]
a = np.array([[float(x) if x != 'nan' else np.nan for x in r[1:]] for r in rows if len(r) > 1])
# the busiest third of the samples (the timed blocks)
if a.size:
    order = np.argsort(a[:, 1] if a.shape[1] > 1 else a[:, 0])[::-1][:max(3, len(a) // 4)]
    print('step %.2f us kernel %.2f us | hwmon columns (freq1, power..., temp) over the loaded samples:' % (d['ms_per_step'] * 1e3, d['roofline']['kernel_ms'] * 1e3), np.round(np.nanmean(a[order], 0), 1), 'max', np.round(np.nanmax(a, 0), 1), 'n', len(a))
else:
    print('step %.2f us kernel %.2f us | no samples' % (d['ms_per_step'] * 1e3, d['roofline']['kernel_ms'] * 1e3))
P
done
