#!/bin/bash
# Shader clock and matrix-core busy fraction of every dispatch of one kernel, from a rocprofv3 --pmc pass (run from the
# repo root on the GPU box; no trace domains in the same run):
#   tools/pmc_clock_busy.sh <kernel name substring> -- <command ...>
#   e.g. tools/pmc_clock_busy.sh wreg128 -- python tools/bench_ares.py --rows 4194304 --n 1024 --variants 2 1
# clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).
pat=$1; shift; [ "$1" = "--" ] && shift
export TMPDIR=/tmp
rm -rf /tmp/pmc_cb
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pmc_cb -o pc -- "$@" > /tmp/pmc_cb.log 2>&1
python - "$pat" <<PY
import glob, sqlite3, sys
db = glob.glob('/tmp/pmc_cb/**/*results.db', recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection "
                   "where kernel_name like ? order by dispatch_id", ('%' + sys.argv[1] + '%',)).fetchall()
cur = {}
for d, n, c, v, dur in rows:
    cur.setdefault(d, {'n': n[:40], 'dur': dur})[c] = v
for d, r in cur.items():
    g, m = r.get('GRBM_GUI_ACTIVE', 0), r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    print('%6d %-40s %8.3f ms  %.2f GHz  MFMA busy %5.1f %%' % (d, r['n'], r['dur'] / 1e6, g / 8 / r['dur'], 100.0 * m / (g / 8 * 1024)))
PY
