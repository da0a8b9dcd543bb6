"""Sum rocprofv3 --pmc counters over the dispatches of vpt_mesh_kernel: python profiles/tools/pmc_sum.py <rocprof output dir>"""
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'vpt_mesh_kernel' in row['Kernel_Name']:
            tot[row['Counter_Name']] += float(row['Counter_Value'])
for k, v in sorted(tot.items()): print(k, v)
