"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace: per kernel name, and per
(kernel name, grid, LDS bytes) so that the line programs of the step -- which share the symbol
`rpde::line_kernel<Cfg>` -- can be told apart.  Usage: rocpd_summary.py results.db out_prefix"""
import sqlite3
import sys


def main(db_path, out):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            f.write('"%s",%d,%.0f,%.1f,%.3f\n' % (r[0], r[1], r[2] * 1e3, r[3] * 1e3, r[4]))
    q = ("select name, grid_x, grid_y, workgroup_x, lds_size, count(*), sum(duration), avg(duration), "
         "min(duration), max(duration) from kernels group by name, grid_x, grid_y, workgroup_x, lds_size "
         "order by sum(duration) desc")
    with open(out + "_dispatch_groups.csv", "w") as f:
        f.write("Name,GridX,GridY,WorkgroupX,LdsBytes,Calls,TotalNs,AverageNs,MinNs,MaxNs\n")
        for r in cur.execute(q):
            f.write('"%s",%d,%d,%d,%d,%d,%d,%.1f,%d,%d\n' % r)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
