"""Print the top kernels of a rocprofv3 results database (rocprofv3 --kernel-trace --stats -d DIR -o NAME)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name, total_calls, average, percentage from top_kernels limit %d" % (int(sys.argv[2]) if len(sys.argv) > 2 else 12)):
    print("%-70s calls %5d  avg %10.1f us  %5.1f%%" % (r[0][:70], r[1], r[2] / 1e3 if r[2] > 1e5 else r[2], r[3]))
