"""median kernel durations of two rocprofv3 kernel traces side by side: python tools/cmp_kernel_trace.py a_kernel_trace.csv b_kernel_trace.csv"""
import csv, collections, statistics, sys
def med(path):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r['Kernel_Name']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    return {k: statistics.median(v[len(v)//2:]) for k,v in d.items()}
a,b=med(sys.argv[1]),med(sys.argv[2])
for k in sorted(a, key=lambda k:-a[k]):
    if k in b and a[k] > 3 and ('GLOBAL__N' in k or 'anonymous' in k): print("%7.1f -> %7.1f  %s"%(a[k],b[k],k[k.find('k_'):][:50]))
