import sqlite3, sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
nsteps=int(sys.argv[2]) if len(sys.argv)>2 else 13
rows=cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, max(vgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 3 desc").fetchall()
tot=sum(r[2] for r in rows)
print(f"total kernel time {tot/1e3:.2f} ms over {nsteps} steps -> {tot/1e3/nsteps:.3f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 25]:
    print(f"{r[0][:80]:80s} n={r[1]:5d} tot={r[2]/1e3:8.2f}ms avg={r[3]:8.1f}us min={r[4]:7.1f} max={r[5]:7.1f} vgpr={r[6]} lds={r[7]} {100*r[2]/tot:5.1f}%")
# GPU busy vs span for the last step-ish
ev=cur.execute("select start,end from kernels order by start").fetchall()
span=(ev[-1][1]-ev[0][0])/1e6; busy=sum(e-s for s,e in ev)/1e6
print(f"span {span:.2f} ms busy {busy:.2f} ms ({100*busy/span:.1f}%)")
