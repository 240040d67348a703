import sys, os
sys.path.insert(0, os.getcwd())
from quadruped_locomotion_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
