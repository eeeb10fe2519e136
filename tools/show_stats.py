import csv, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/*_kernel_stats.csv')):
    print('==', f.split('/')[-1])
    for r in list(csv.DictReader(open(f)))[:7]:
        print('  ', r['Name'][:80].replace('(anonymous namespace)::', '').ljust(70), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:10.1f} us", r['Percentage'])
