import sqlite3,glob,sys
db=sqlite3.connect(glob.glob(sys.argv[1]+'/*.db')[0])
c=db.cursor()
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=c.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
for n,d in rows:
    if 'rma_' in n: print(n[:40], d/1000.)
