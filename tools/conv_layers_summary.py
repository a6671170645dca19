import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'spconv' in r['Kernel_Name'] or 'conv2d_tile' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = open(sys.argv[2]).read().split('\n')
names = [n for n in names if '->' in n]
for i in range(0, len(rows), 10):
    grp = rows[i:i + 10]
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in grp)
    r = grp[0]
    kn = r['Kernel_Name'].split('(')[0].replace('void (anonymous namespace)::', '')
    print(f"{names[i // 10] if i // 10 < len(names) else '?':22s} {kn:38s} grid {int(r['Grid_Size_X']) // 256}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} vgpr {r['VGPR_Count']:>3s}  median {d[len(d) // 2]:6.1f} us  min {d[0]:6.1f}")
