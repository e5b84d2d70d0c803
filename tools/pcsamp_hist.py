"""Aggregate a rocprofv3 PC-sampling CSV per kernel and instruction (run on the GPU box: the raw file is too large to bring back)."""
import csv, glob, os, sys, collections
src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
kt = glob.glob(src + '/**/*kernel_trace.csv', recursive=True)
names = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        names[r.get('Dispatch_Id')] = r.get('Kernel_Name', '?').replace('void ', '').replace('sjmi::', '').split('(')[0]
files = [f for f in glob.glob(src + '/**/*pc_sampling*.csv', recursive=True)]
print('files', files)
hist = collections.defaultdict(collections.Counter)
tot = collections.Counter()
head = []
for f in files:
    with open(f) as fh:
        rd = csv.DictReader(fh)
        for i, r in enumerate(rd):
            if i < 30: head.append(r)
            k = names.get(r.get('Dispatch_Id'), '?')
            key = (r.get('Instruction', ''), r.get('Instruction_Comment', ''))
            extra = tuple((c, r[c]) for c in r if c.startswith('Stall') or c in ('Wave_Issued', 'Instruction_Type', 'Stall_Reason'))
            hist[k][key + (extra if False else ())] += 1
            tot[k] += 1
            for c, v in extra: hist[k + '#' + c][v] += 1
with open(out + '/head.txt', 'w') as fh:
    for r in head: fh.write(repr(r) + '\n')
with open(out + '/hist.txt', 'w') as fh:
    for k, n in tot.most_common():
        fh.write('== %s: %d samples\n' % (k, n))
        for (key, c) in hist[k].most_common(600):
            fh.write('%7d %5.2f%%  %s\n' % (c, 100.0 * c / n, ' | '.join(map(str, key))))
    for k in hist:
        if '#' in k:
            fh.write('== %s\n' % k)
            for key, c in hist[k].most_common(40): fh.write('%9d  %s\n' % (c, key))
print({k: v for k, v in tot.most_common(12)})
