"""Print the kernels of one headline step (start, duration, grid, queue) from a rocprofv3
--kernel-trace database of bench.py, e.g.

    rocprofv3 --kernel-trace -d out -o bench -- python3 bench.py --no-cpu-baseline --no-also
    python3 tools/step_timeline.py out/bench_results.db [steps back from the end] [sweep kernel name part]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    which = sys.argv[3] if len(sys.argv) > 3 else "sweep_tile_kernel<16,"     # the headline shape's sweep kernel
    rows = list(db.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x,queue_id from kernels order by start"))
    sweeps = [i for i, r in enumerate(rows) if which in r[0]]
    lo, hi = sweeps[-back - 1] + 1, sweeps[-back] + 1
    t0 = rows[lo][1]
    for r in rows[lo:hi + 3]:
        name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
        print("%8.1f %7.1f  grid=(%d,%d,%d)/%d q%s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4], r[5],
                                                         r[6], r[7], name[:60]))
    print("step period (sweep start to sweep start): %.1f us" % ((rows[sweeps[-back]][1] - rows[sweeps[-back - 1]][1]) / 1e3))


if __name__ == "__main__":
    main()
