"""bench.py's 512^3 step three times without the CPU leg and the secondary configurations (the script of PMC passes)."""
import sys, runpy
sys.argv = ["bench.py", "--mesh", "512", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-secondary"]
runpy.run_path("bench.py", run_name="__main__")
