"""Per-round kernel durations of the LAST build in a rocprofv3 kernel trace (csv): python tools/hpr_round_table.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
last = max(i for i, n in enumerate(names) if 'k_load' in n)
rows = rows[last:]
t0 = int(rows[0]['Start_Timestamp'])
rnd, cur = [], {}
other = {}
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('hull::', '').replace('void ', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000
    if 'owner_claim' in n:
        if cur: rnd.append(cur)
        cur = {'claim': d, 't': (int(r['Start_Timestamp']) - t0) / 1000}
    elif cur and n == 'k_reassign_only' and n in cur:
        cur['reassign_again'] = cur.get('reassign_again', 0) + d   # (TOHIP_HULL_SPLIT_LINK=2: a second pass that moves nobody)
    elif cur and n in ('k_new_faces', 'k_link_reassign', 'k_link_only', 'k_reassign_only', 'k_round_tail', 'k_accept'):
        cur[n] = cur.get(n, 0) + d
    else:
        other[n[:50]] = other.get(n[:50], 0) + d
if cur: rnd.append(cur)
keys = ['claim', 'k_new_faces', 'k_link_reassign', 'k_link_only', 'k_reassign_only', 'reassign_again', 'k_round_tail']
print('round   t_us ' + ' '.join(f'{k[-12:]:>12s}' for k in keys))
for i, c in enumerate(rnd):
    print(f'{i:5d} {c["t"]:7.0f} ' + ' '.join(f'{c.get(k, 0):12.1f}' for k in keys))
print('sum         ' + ' '.join(f'{sum(c.get(k, 0) for c in rnd):12.1f}' for k in keys))
for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:25]:
    print(f'{v:9.1f} us  {k}')
