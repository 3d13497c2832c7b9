"""bench.py on an A/B build of the library (variants/NAME.so from tools/ab_variants.py build): python tools/ab_bench.py variants/NAME.so [bench.py arguments]"""
import sys, runpy
sys.path.insert(0, ".")
from livelyspeaker_amd import _lib
_lib.use_library(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
