"""Diagnostic: python tools/bench_with_halos.py <fwd_halo> <bwd_halo> <bench args...> (tiled advection windows)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops
ops.ADVECT_FLAGS = ops.advect_flags(halo=int(sys.argv[1]), halo_bwd=int(sys.argv[2]))
sys.argv = ["bench.py"] + sys.argv[3:]
import bench
bench.main()
