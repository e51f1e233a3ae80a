export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pc -o pc -- python $GRAFT_REPO_ROOT/tools/bench_ares.py --rows 4194304 --n 1024 --variants 2 50 34 3 > /tmp/pc.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pc/**/*results.db', recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%wreg128%' order by dispatch_id").fetchall()
cur = {}
for d, n, c, v, dur in rows:
    cur.setdefault(d, {'n': n[:30], 'dur': dur})[c] = v
for d, r in cur.items():
    g = r.get('GRBM_GUI_ACTIVE', 0); m = r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    print(d, r['n'], 'dur %.3f ms' % (r['dur'] / 1e6), 'clk %.2f GHz' % (g / 8 / r['dur']), 'mfma busy %.1f %%' % (100.0 * m / (g / 8 * 1024)))
PY
