"""Summarise a rocprofv3 rocpd (SQLite) result into a kernel-stats text table.

    python tools/rocpd_summary.py gpurun_out/<dir>/<name>_results.db > profiles/<round>_<name>_kernel_stats.txt

Equivalent of `rocprofv3 --stats` CSV output (kernel name, calls, total/avg/min/max duration in ns, %).
"""
import sqlite3
import sys


def main(path, pmc=False):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print("# name | calls | total_ns | avg_ns | min_ns | max_ns | pct | vgpr | agpr | sgpr | lds | grid_x | wg_x")
    for r in rows:
        name = r[0] if len(r[0]) < 110 else r[0][:107] + '...'
        print(f"{name} | {r[1]} | {r[2]} | {r[3]:.0f} | {r[4]} | {r[5]} | {100.0 * r[2] / total:.2f} | "
              f"{r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]}")
    if pmc:
        try:
            q = cur.execute("select * from counters_collection limit 1")
            cols = [d[0] for d in q.description]
            print('# counters_collection columns:', cols)
            rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                               "group by kernel_name, counter_name order by kernel_name").fetchall()
            for r in rows:
                print(' | '.join(str(x) for x in r))
        except Exception as e:  # noqa
            print('# no counters:', e)


if __name__ == '__main__':
    main(sys.argv[1], pmc=len(sys.argv) > 2)
