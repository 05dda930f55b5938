#!/usr/bin/env python3
"""rocprofv3 rocpd database -> compact kernel table (name, stream, start, end; gzip CSV) + per-kernel statistics CSV.
   python tools/gpu/slim_trace.py trace_results.db out_dir"""
import csv
import gzip
import os
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
out = sys.argv[2]
with gzip.open(os.path.join(out, 'kernels.csv.gz'), 'wt', newline='') as f:
    w = csv.writer(f)
    w.writerow(['name', 'stream', 'start', 'end'])
    for r in db.execute('select name, stream_id, start, end from kernels order by start'):
        w.writerow(r)
with open(os.path.join(out, 'kernel_stats.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['name', 'calls', 'total_us', 'avg_us', 'pct'])          # rocpd top_kernels reports microseconds
    for r in db.execute('select name, total_calls, total_duration, average, percentage from top_kernels'):
        w.writerow(r)
