"""A/B of the fused GlobalBias projection: python tools/ab_bias_proj.py <0|1> <bench args...>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import paradis_model_amd.model.blocks as blocks
blocks.FUSE_BIAS_PROJECTION = bool(int(sys.argv[1]))
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
