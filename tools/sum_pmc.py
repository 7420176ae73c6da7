import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r['Kernel_Name'][:40], r['Counter_Name'])
            acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(d, k, 'sum', v, 'dispatches', n, 'per dispatch', v / n)
