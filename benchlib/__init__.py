"""The pieces of ``bench.py`` (launcher, workloads, the timed chain run, post-run legs, CPU baselines)."""
