"""Parts of bench.py (the repo-root benchmark the driver runs): launch = command line + rank processes, counters = rocprofv3
counter passes, workloads = the three Step classes, report = the JSON line and its roofline arithmetic, evidence = untimed accuracy
checks.  bench.py itself keeps the timed loops and the CPU-baseline leg (the only code that imports oracle/)."""
