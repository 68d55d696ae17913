#!/usr/bin/env python3
"""timed.py OUTFILE cmd args...: runs the command (stdin/stdout/stderr passed through), writes "<seconds> s wall, <KiB> KiB peak RSS"
(the child's own ru_maxrss, wait4) to OUTFILE and exits with the command's status.  (/usr/bin/time is not in the image.)"""
import os
import subprocess
import sys
import time

t = time.perf_counter()
p = subprocess.Popen(sys.argv[2:])
_, status, ru = os.wait4(p.pid, 0)
el = time.perf_counter() - t
open(sys.argv[1], "w").write("%.3f s wall, %d KiB peak RSS\n" % (el, ru.ru_maxrss))
sys.exit(status >> 8 if status & 0xff == 0 else 128 + (status & 0x7f))
