"""Summarise a rocprofv3 kernel trace of bench.py: per-kernel mean duration and GPU idle gaps over the last env-steps."""
import sys, glob
import pandas as pd
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
df = pd.read_csv(f).sort_values('Start_Timestamp')
df['k'] = df.Kernel_Name.str.extract(r'(k_[a-z_]+)')[0].fillna('other')
main = df[df.k.isin(['k_kinematics', 'k_cull', 'k_narrow', 'k_solve_mf', 'k_solve_g', 'k_solve', 'k_collide'])]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600          # substeps to look at (from the end)
nk = main.k.nunique()
tail = main.tail(n * nk)
dur = (tail.End_Timestamp - tail.Start_Timestamp)
print((dur.groupby(tail.k).mean() / 1e3).round(1).to_string(), '(us mean)')
span = (tail.End_Timestamp.max() - tail.Start_Timestamp.min()) / 1e3
busy = dur.sum() / 1e3
print(f'span {span / n:.1f} us/substep, kernels busy {busy / n:.1f} us/substep, idle {100 * (1 - busy / span):.1f} %')
