"""one step of a cold leg from rocprofv3's kernel trace: start, idle gap in front, duration of every kernel (small rocPRIM kernels folded)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'row_classes' in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
fold_n = fold_gap = fold_dur = 0
for r in rows[i0:]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    name = r['Kernel_Name']
    small = ('rocprim' in name or 'rocclr' in name) and e - s < 30000
    if small:
        fold_n += 1; fold_gap += s - prev_end; fold_dur += e - s
    else:
        if fold_n:
            print(f"{'':9}     {fold_n:3d} small kernels: gaps {fold_gap/1e3:7.1f}  dur {fold_dur/1e3:8.1f}")
            fold_n = fold_gap = fold_dur = 0
        print(f"{(s-t0)/1e3:9.1f} us  gap {(s-prev_end)/1e3:7.1f}  dur {(e-s)/1e3:8.1f}  {name[:70]}")
    prev_end = e
