import sqlite3, glob, sys
db = glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True)[0]
con = sqlite3.connect(db); cur = con.cursor()
rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3 from kernels group by name order by 3 desc limit %d" % int(sys.argv[2] if len(sys.argv)>2 else 18)).fetchall()
for r in rows: print("%-100s n=%5d total %10.1f us avg %9.1f min %9.1f" % (r[0][:100], r[1], r[2], r[3], r[4]))
