import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
rows=c.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by count(*) desc").fetchall()
for r in rows[:8]: print(r[0][:60], r[1], round(r[2]/1e3,2), round(r[3]/1e3,2))
rows=c.execute("select name,start,end from kernels order by start").fetchall()
seq=rows[-14:]
t0=seq[0][1]
for n,s,e in seq: print(round((s-t0)/1e3,1), round((e-s)/1e3,1), n[:50])
