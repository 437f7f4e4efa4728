#!/usr/bin/env python3
"""Per-round kernel time table of a hull build from a rocprofv3 sqlite database (rocprofv3 --kernel-trace -d DIR):
python tools/hull_round_table.py DIR/x_results.db [call_index]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id "
                        "where s.kernel_name like '%hull%' order by d.start"))
inits = [i for i, r in enumerate(rows) if 'k_init' in r[0]]
call = int(sys.argv[2]) if len(sys.argv) > 2 else len(inits) - 1
rows = rows[inits[call]:(inits[call + 1] if call + 1 < len(inits) else None)]
rnd = -1
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
tot = collections.defaultdict(lambda: [0.0, 0])
for n, s, d in rows:
    k = n.split('hull')[1].lstrip('0123456789')[:12]
    if 'round_reset' in n: rnd += 1
    tot[k][0] += d / 1e3; tot[k][1] += 1
    if rnd >= 0: per[rnd][k][0] += d / 1e3; per[rnd][k][1] += 1
print("rounds", rnd + 1, "span_ms", (rows[-1][1] - rows[0][1]) / 1e6)
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:14s} {t/1e3:8.2f} ms  {c:6d} launches")
step = max(1, (rnd + 1) // 16)
for r in range(0, rnd + 1, step):
    print(r, {k[2:10]: (round(v[0], 1), v[1]) for k, v in per[r].items() if v[0] > 8})
