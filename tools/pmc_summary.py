"""summarise a rocprofv3 --pmc results DB: per kernel (name, workgroups) average of every counter + duration"""
import sqlite3, sys, json
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, grid_size_x/workgroup_size_x, counter_name, avg(value), avg(end-start), count(*) "
                 "from counters_collection where kernel_name like '%qmm%' or kernel_name like '%paged%' or kernel_name like '%argmax%' "
                 "group by kernel_name, grid_size_x, counter_name").fetchall()
d = defaultdict(dict)
for r in rows:
    k = f"{r[0][:48]} wgs={r[1]}"
    d[k][r[2]] = r[3]
    d[k]["dur_us"] = r[4] / 1e3
    d[k]["launches"] = r[5]
print(json.dumps(d, indent=1))
