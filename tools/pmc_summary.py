import sqlite3, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*.db')[0]
db = sqlite3.connect(f); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
sfx = [t for t in tabs if t.startswith('rocpd_pmc_event')][0][len('rocpd_pmc_event'):]
q = f"""select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from rocpd_pmc_event{sfx} e
 join rocpd_info_pmc{sfx} p on e.pmc_id=p.id
 join rocpd_kernel_dispatch{sfx} d on e.event_id=d.event_id
 join rocpd_info_kernel_symbol{sfx} s on d.kernel_id=s.id group by s.kernel_name,p.name"""
res = collections.defaultdict(dict)
for k, c, v, n in cur.execute(q): res[k.split('(')[0][:48]][c] = v / max(n, 1)
for k, d in res.items():
    w = d.get('SQ_WAVES', 0) or 1
    print(k, {c: (round(v / w, 1) if c != 'SQ_WAVES' else round(v)) for c, v in sorted(d.items())})
