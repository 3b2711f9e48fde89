import sqlite3, sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
pat=sys.argv[2] if len(sys.argv)>2 else '%'
cols=[r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows=cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name order by kernel_name", (pat,)).fetchall()
for r in rows: print(f"{r[0][:50]:50s} {r[1]:28s} avg={r[2]:16.1f} n={r[3]}")
